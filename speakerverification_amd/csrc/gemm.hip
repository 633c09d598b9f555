// gemm.hip — LDS-tiled MFMA GEMM for the TDNN / Res2Net / MFA / ASP convolutions (gfx950).
//
// One kernel template covers both compute dtypes because the byte geometry is identical:
// a K-step is 128 bytes per row (64 bf16 or 32 fp32), moved as 16-byte chunks.
//   tile 128 x 128, 256 threads = 4 waves as 2 (M) x 2 (N), each wave 64 x 64 = 2 x 2 MFMA 32x32 tiles
//   bf16: v_mfma_f32_32x32x16_bf16, fp32: v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, parity path)
//   LDS: double-buffered A and B tiles (4 x 16 KiB), register-staged prefetch of tile k+1 under
//        the MFMAs of tile k, one barrier per K-step; 16-byte chunk XOR swizzle
//        chunk' = chunk ^ ((row >> 1) & 7) makes every ds_read_b128 fragment read conflict-free.
//   A operand = activations (rows m, frame-major), B operand = packed weights [N][K]:
//        the MFMA C layout then has n on the lane, so bias / BN scale / shift are per-lane scalars.
//   grid: XCD-aware — each XCD owns a contiguous band of M tiles and walks it in groups of 8 M-tiles
//        x all N-tiles (m fastest), so concurrently resident workgroups share A and W panels in that
//        XCD's L2.
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace svhip {

namespace {

constexpr int BM = GEMM_BM, BN = GEMM_BN;
constexpr int ROWB = GEMM_BK_BYTES;            // bytes per tile row
constexpr int TILE_BYTES = BM * ROWB;          // 16 KiB
constexpr int GROUP_M = 8;

__device__ __forceinline__ int swz(int row, int chunk) {
    return row * ROWB + ((chunk ^ ((row >> 1) & 7)) << 4);
}

template <typename T> struct MmaTraits;
template <> struct MmaTraits<float> {
    static constexpr int EPC = 4;              // elements per 16-byte chunk
    static constexpr int BK = 32;
    typedef f32x4 chunk_t;
    static __device__ __forceinline__ void mma(const chunk_t& a, const chunk_t& b, f32x16& c) {
#pragma unroll
        for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], c, 0, 0, 0);
    }
    static __device__ __forceinline__ chunk_t zero() { return chunk_t{0.f, 0.f, 0.f, 0.f}; }
    static __device__ __forceinline__ chunk_t add(const chunk_t& a, const chunk_t& b) { return a + b; }
};
template <> struct MmaTraits<bf16_t> {
    static constexpr int EPC = 8;
    static constexpr int BK = 64;
    typedef bf16x8 chunk_t;
    static __device__ __forceinline__ void mma(const chunk_t& a, const chunk_t& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ chunk_t zero() {
        chunk_t z;
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = static_cast<bf16_t>(0.0f);
        return z;
    }
    static __device__ __forceinline__ chunk_t add(const chunk_t& a, const chunk_t& b) {
        chunk_t r;
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = static_cast<bf16_t>(static_cast<float>(a[j]) + static_cast<float>(b[j]));
        return r;
    }
};

template <> struct MmaTraits<f16_t> {         // SVHIP_F16 handles (RawNet2)
    static constexpr int EPC = 8;
    static constexpr int BK = 64;
    typedef f16x8 chunk_t;
    static __device__ __forceinline__ void mma(const chunk_t& a, const chunk_t& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ chunk_t zero() {
        chunk_t z;
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = static_cast<f16_t>(0.0f);
        return z;
    }
    static __device__ __forceinline__ chunk_t add(const chunk_t& a, const chunk_t& b) {
        chunk_t r;
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = static_cast<f16_t>(static_cast<float>(a[j]) + static_cast<float>(b[j]));
        return r;
    }
};

// activation pairs used by the models, fixed at compile time (a runtime switch inlined 128 times
// made the epilogue 30k instructions long and blew the instruction cache)
enum Epi : int { EPI_NONE = 0, EPI_RELU = 1, EPI_GELU = 2, EPI_RELU_TANH = 3, EPI_LRELU03 = 4, EPI_BN_LRELU03 = 5, EPI_LRELU001 = 6 };

// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7) on the fast exp / rcp units: bf16 path only
__device__ __forceinline__ float gelu_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float e = __expf(-z * z);
    const float erf_abs = fmaf(-poly * t, e, 1.0f);          // erf(|x|/sqrt2)
    const float ax = fabsf(x);
    // 0.5*x*(1+erf(x/sqrt2)) = 0.5*(x + |x|*erf_abs)
    return 0.5f * fmaf(ax, erf_abs, x);
}

template <typename T, int EPI>
__device__ __forceinline__ float epilogue_act1(float v) {
    if (EPI == EPI_RELU || EPI == EPI_RELU_TANH) return fmaxf(v, 0.0f);
    if (EPI == EPI_GELU) return sizeof(T) == 2 ? gelu_fast(v) : 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    if (EPI == EPI_LRELU03) return v > 0.0f ? v : 0.3f * v;
    if (EPI == EPI_LRELU001) return v > 0.0f ? v : 0.01f * v;
    return v;
}

// fp32 value pair -> bf16 hi / lo parts (x = hi + lo up to 2^-17 relative): eight k-values of one fragment row
// (planes of x3_t, common.h: IEEE half since round 4 — 2^-23 relative while lo is a normal half; the fragments travel as raw 16-byte vectors)
__device__ __forceinline__ void split_hi_lo(const f32x4& a, const f32x4& b, bf16x8& hi, bf16x8& lo) {
    x3x8_t h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const x3_t ha = x3_hi(a[e]), hb = x3_hi(b[e]);
        h[e] = ha; h[4 + e] = hb;
        l[e] = x3_lo(a[e], ha);
        l[4 + e] = x3_lo(b[e], hb);
    }
    hi = __builtin_bit_cast(bf16x8, h);
    lo = __builtin_bit_cast(bf16x8, l);
}

// X3 (T = float only, SVHIP_F32X3 handles): every product as three bf16 MFMAs on hi / lo-split fragments (see gemm_pw.hip)
template <typename T, bool CONV, bool HAS_A2, int EPI, bool X3 = false>
__global__ __launch_bounds__(256) void gemm_kernel(GemmParams p) {
    typedef MmaTraits<T> TR;
    typedef typename TR::chunk_t chunk_t;
    constexpr int EPC = TR::EPC;
    constexpr int BK = TR::BK;

    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][A 16K | B 16K]

    // ---- XCD-aware tile mapping --------------------------------------------------------------
    const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
    const int nwg = ntm * ntn;
    int id = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = id & 7;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
    }
    const int grp = id / (GROUP_M * ntn);
    const int within = id - grp * (GROUP_M * ntn);
    const int gm = min(GROUP_M, ntm - grp * GROUP_M);
    const int tile_m = grp * GROUP_M + within % gm;
    const int tile_n = within / gm;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // ---- per-thread staging geometry: 4 A chunks + 4 B chunks per K-step ----------------------
    const int lc = tid & 7;                  // chunk column 0..7
    const int lr = tid >> 3;                 // row 0..31 (+32*i)
    const T* __restrict__ Ap = reinterpret_cast<const T*>(p.A);
    const T* __restrict__ A2p = reinterpret_cast<const T*>(p.A2);
    const T* __restrict__ Wp = reinterpret_cast<const T*>(p.W);

    int rowA[4];        // clamped global row
    int uttA[4], tA[4]; // utterance base row and frame index (CONV only)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = min(m0 + lr + 32 * i, p.M - 1);
        rowA[i] = m;
        if (CONV) {
            int b = m / p.T;
            uttA[i] = b * p.T;
            tA[i] = m - b * p.T;
        }
    }
    int64_t wrow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) wrow[i] = (int64_t)min(n0 + lr + 32 * i, p.Wrows - 1) * p.Kp;

    chunk_t ra[4], rb[4];

    auto load_tile = [&](int kt) {
        const int k = kt * BK + lc * EPC;
        const bool kvalid = k < p.K;
        int tap_off = 0, cc = k;
        if (CONV) {
            const int tap = k / p.cin;
            cc = k - tap * p.cin;
            tap_off = (tap - (p.taps >> 1)) * p.dil;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            chunk_t v = TR::zero();
            if (kvalid) {
                if (CONV) {
                    int tt = tA[i] + tap_off;
                    bool ok = true;
                    if (p.pad_mode == PAD_REFLECT) tt = reflect_idx(tt, p.T);
                    else ok = (tt >= 0) && (tt < p.T);
                    if (ok) {
                        const int64_t src = (int64_t)(uttA[i] + tt);
                        v = *reinterpret_cast<const chunk_t*>(Ap + src * p.lda + cc);
                        if (HAS_A2) v = TR::add(v, *reinterpret_cast<const chunk_t*>(A2p + src * p.lda2 + cc));
                    }
                } else {
                    v = *reinterpret_cast<const chunk_t*>(Ap + (int64_t)rowA[i] * p.lda + k);
                    if (HAS_A2) v = TR::add(v, *reinterpret_cast<const chunk_t*>(A2p + (int64_t)rowA[i] * p.lda2 + k));
                }
            }
            ra[i] = v;
            rb[i] = *reinterpret_cast<const chunk_t*>(Wp + wrow[i] + kt * BK + lc * EPC);
        }
    };
    auto store_tile = [&](int buf) {
        char* As = smem + buf * (2 * TILE_BYTES);
        char* Bs = As + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = lr + 32 * i;
            *reinterpret_cast<chunk_t*>(As + swz(r, lc)) = ra[i];
            *reinterpret_cast<chunk_t*>(Bs + swz(r, lc)) = rb[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int nk = p.Kp / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();

    const int fr = lane & 31, fh = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        const char* As = smem + buf * (2 * TILE_BYTES);
        const char* Bs = As + TILE_BYTES;
        if constexpr (X3) {
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {         // chunks (4pr + fh, 4pr + 2 + fh) of a row = this lane's 8 of the step's 16 k-values
                bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    split_hi_lo(*reinterpret_cast<const f32x4*>(As + swz(wm * 64 + i * 32 + fr, 4 * pr + fh)),
                                *reinterpret_cast<const f32x4*>(As + swz(wm * 64 + i * 32 + fr, 4 * pr + 2 + fh)), ah[i], al[i]);
                    split_hi_lo(*reinterpret_cast<const f32x4*>(Bs + swz(wn * 64 + i * 32 + fr, 4 * pr + fh)),
                                *reinterpret_cast<const f32x4*>(Bs + swz(wn * 64 + i * 32 + fr, 4 * pr + 2 + fh)), bh[i], bl[i]);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[i][j] = X3H::mfma32(al[i], bh[j], acc[i][j]);
                        acc[i][j] = X3H::mfma32(ah[i], bl[j], acc[i][j]);
                        acc[i][j] = X3H::mfma32(ah[i], bh[j], acc[i][j]);
                    }
            }
        } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            chunk_t af[2], bfr[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = *reinterpret_cast<const chunk_t*>(As + swz(wm * 64 + i * 32 + fr, 2 * s + fh));
                bfr[i] = *reinterpret_cast<const chunk_t*>(Bs + swz(wn * 64 + i * 32 + fr, 2 * s + fh));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) TR::mma(af[i], bfr[j], acc[i][j]);
        }
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: bias -> act1 -> BN affine -> act2 -> store --------------------------------
    const bool out_f32 = (sizeof(T) == 4) || p.out_f32;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + fr;
        const bool nok = n < p.N;
        const float bias = (p.bias && nok) ? p.bias[n] : 0.0f;
        const float sc = (p.scale && nok) ? p.scale[n] : 1.0f;
        const float sh = (p.shift && nok) ? p.shift[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (nok && m < p.M) {
                    float v = acc[i][j][r] + bias;
                    if (p.bias_utt) v += p.bias_utt[(int64_t)(m / p.T) * p.ld_bu + n];
                    v = epilogue_act1<T, EPI>(v);
                    v = fmaf(v, sc, sh);
                    if (EPI == EPI_RELU_TANH) v = tanhf(v);
                    if (EPI == EPI_BN_LRELU03) v = v > 0.0f ? v : 0.3f * v;
                    if (p.R) v += to_f32<T>(reinterpret_cast<const T*>(p.R)[(int64_t)m * p.ldr + n]);
                    if (out_f32) reinterpret_cast<float*>(p.Y)[(int64_t)m * p.ldy + n] = v;
                    else {
                        typedef typename std::conditional<sizeof(T) == 2, T, bf16_t>::type OT;
                        reinterpret_cast<OT*>(p.Y)[(int64_t)m * p.ldy + n] = static_cast<OT>(v);
                    }
                }
            }
        }
    }
}

template <typename T, bool CONV, bool HAS_A2, int EPI, bool X3>
hipError_t launch_inst_x(const GemmParams& p, hipStream_t stream) {
    const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
    dim3 grid(ntm * ntn), block(256);
    const size_t lds = 4 * TILE_BYTES;
    static DeviceOnce attr;         // 64 KiB of dynamic LDS per workgroup, raised once per device
    if (hipError_t e = set_max_dynamic_lds(attr, reinterpret_cast<const void*>(gemm_kernel<T, CONV, HAS_A2, EPI, X3>), (int)lds)) return e;
    hipLaunchKernelGGL((gemm_kernel<T, CONV, HAS_A2, EPI, X3>), grid, block, lds, stream, p);
    return hipGetLastError();
}

template <typename T, bool CONV, bool HAS_A2, int EPI>
hipError_t launch_inst(const GemmParams& p, hipStream_t stream) {
    if constexpr (sizeof(T) == 4) {
        if (p.x3) return launch_inst_x<T, CONV, HAS_A2, EPI, true>(p, stream);
    }
    return launch_inst_x<T, CONV, HAS_A2, EPI, false>(p, stream);
}

template <typename T, int EPI>
hipError_t launch_epi(const GemmParams& p, hipStream_t stream) {
    const bool conv = p.taps > 1;
    if (conv) return p.A2 ? launch_inst<T, true, true, EPI>(p, stream) : launch_inst<T, true, false, EPI>(p, stream);
    if (p.A2) return hipErrorInvalidValue;       // pointwise + A2 is not used by any model
    return launch_inst<T, false, false, EPI>(p, stream);
}

template <typename T>
hipError_t launch_t(const GemmParams& p, hipStream_t stream) {
    if (p.act1 == ACT_NONE && p.act2 == ACT_NONE) return launch_epi<T, EPI_NONE>(p, stream);
    if constexpr (!std::is_same<T, f16_t>::value) {      // (fp16: RawNet2's epilogues only)
    if (p.act1 == ACT_RELU && p.act2 == ACT_NONE) return launch_epi<T, EPI_RELU>(p, stream);
    if (p.act1 == ACT_GELU && p.act2 == ACT_NONE) return launch_epi<T, EPI_GELU>(p, stream);
    if (p.act1 == ACT_RELU && p.act2 == ACT_TANH) return launch_epi<T, EPI_RELU_TANH>(p, stream);
    }
    if (p.act1 == ACT_LRELU03 && p.act2 == ACT_NONE) return launch_epi<T, EPI_LRELU03>(p, stream);
    if (p.act1 == ACT_NONE && p.act2 == ACT_LRELU03) return launch_epi<T, EPI_BN_LRELU03>(p, stream);
    if (p.act1 == ACT_LRELU001 && p.act2 == ACT_NONE) return launch_epi<T, EPI_LRELU001>(p, stream);
    return hipErrorInvalidValue;
}

}  // namespace

GemmRoute gemm_route(const GemmParams& p, bool bf16) {
#ifdef SVHIP_GEMM_DEBUG
    const bool skip_pw2 = (p.debug & 8) != 0;               // tools/gemm_bench: A/B the 256 x 256 kernel against gemm_pw
    const bool no_narrow = (p.debug & 64) != 0;
    const bool no_pw3 = (p.debug & 4096) != 0;              // A/B the persistent kernel against the per-tile one
#else
    const bool skip_pw2 = false, no_narrow = false, no_pw3 = false;
#endif
    const bool pw = gemm_pw_supported(p, bf16);
    if (bf16 && p.taps > 1 && !p.cv_off && gemm_pw3cv16_supported(p)) return ROUTE_PW3CV;
    if (bf16 && !p.n128_off && gemm_n128_supported(p)) return ROUTE_N128;
    if (!skip_pw2 && gemm_pw2_supported(p, bf16)) {
        const int tiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
        const bool pw3 = !no_pw3 && gemm_pw3_supported(p, bf16);
        // (a capped persistent grid — the test switch SVHIP_PW3_CUS — also takes the small grids that would go to the narrow tile)
        // (round 5: the persistent kernel walks such a grid as column halves of its 256 x 256 tiles — pw3_half_off keeps the narrow tile)
        if (pw && !no_narrow && !p.colsum && 2 * tiles <= p.num_cu && !(pw3 && (pw3_grid_cap(p) < p.num_cu || p.tail_split))) return ROUTE_PW_NARROW;
        if (pw3) return ROUTE_PW3;
        return ROUTE_PW2;
    }
    return pw ? ROUTE_PW : ROUTE_GENERIC;
}

int gemm_colsum_groups(const GemmParams& p, bool bf16) { return gemm_route(p, bf16) == ROUTE_PW3 ? 2 : 8; }

hipError_t launch_gemm(const GemmParams& p, bool bf16, hipStream_t stream) {
    // host-side shape contract (checked before any launch: a bad shape must not reach the GPU)
    const int epc = bf16 ? 8 : 4;
    const int bk = gemm_bk(bf16);
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || p.Kp % bk != 0 || p.Kp < p.K || p.Wrows < p.N) return hipErrorInvalidValue;
    if (p.lda % epc != 0 || (p.A2 && p.lda2 % epc != 0)) return hipErrorInvalidValue;
    if (p.taps > 1) {
        if (p.cin % epc != 0 || p.taps * p.cin != p.K || p.T <= 0 || p.M % p.T != 0) return hipErrorInvalidValue;
        if (p.pad_mode == PAD_REFLECT && (p.taps / 2) * p.dil >= p.T) return hipErrorInvalidValue;
    } else if (p.K % epc != 0) {
        return hipErrorInvalidValue;
    }
    if (p.bias_utt && (p.T <= 0)) return hipErrorInvalidValue;
    if (p.A3 && gemm_route(p, bf16) != ROUTE_PW2) return hipErrorInvalidValue;       // (only the 256 x 256 kernel reads a second K segment)
    switch (gemm_route(p, bf16)) {
        case ROUTE_PW2: return launch_gemm_pw2(p, stream);
        case ROUTE_PW3: return (p.pw4 && gemm_pw4_supported(p, bf16)) ? launch_gemm_pw4(p, stream) : launch_gemm_pw3(p, stream);
        case ROUTE_PW3CV: return launch_gemm_pw3cv16(p, stream);
        case ROUTE_N128: return launch_gemm_n128(p, stream);
        case ROUTE_PW: return launch_gemm_pw(p, bf16, stream);
        case ROUTE_PW_NARROW: return launch_gemm_pw(p, bf16, stream, true);
        default: break;
    }
    if (bf16 && p.f16) return p.x3 ? hipErrorInvalidValue : launch_t<f16_t>(p, stream);
    return bf16 ? launch_t<bf16_t>(p, stream) : launch_t<float>(p, stream);
}

}  // namespace svhip
