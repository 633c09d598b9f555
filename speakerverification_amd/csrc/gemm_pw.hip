// gemm_pw.hip — the dominant kernel: pointwise (k=1) convolution as an MFMA GEMM, 256 x 128 x (128 B) tiles.
//
//   Y[m, n] = epi( sum_k A[m, k] * W[n, k] ),  A frame-major activations (lda), W packed [Np][Kp].
//
// Structure (gfx950):
//   * 512 threads = 8 waves as 4 (M) x 2 (N), each wave 64 x 64 = 2 x 2 MFMA 32x32 tiles,
//     bf16: v_mfma_f32_32x32x16_bf16, fp32: v_mfma_f32_32x32x2_f32; one workgroup per CU.
//   * operands go HBM/L2 -> LDS with global_load_lds_dwordx4 (no VGPR staging): 3-stage ring of
//     (A 32 KiB + W 16 KiB); loads for K-step k+2 are issued before the MFMAs of step k and are
//     retired with a COUNTED s_waitcnt vmcnt(6) + raw s_barrier, so two K-steps of loads stay in
//     flight across barriers.  LDS is written lane-linear by the DMA, so the bank-conflict swizzle
//     (16-byte chunk ^ ((row >> 1) & 7)) is applied on the per-lane SOURCE address and again on
//     the fragment reads.
//   * the weights are the MFMA "A" operand and the activations the "B" operand: the accumulator
//     then holds 4 consecutive output channels per register group for one frame per lane, so the
//     epilogue (bias -> activation -> BatchNorm affine [-> tanh]) packs 4 outputs, writes them to
//     an XOR-swizzled LDS image of the 256 x 128 output tile and the tile leaves the CU as whole
//     256-byte rows in 16-byte stores.
//   * XCD-aware grid: each XCD walks a contiguous band of the tile space in groups of 4 M-tiles x
//     all N-tiles, so co-resident workgroups share A and W panels in that XCD's L2.
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace svhip {

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

#ifdef SVHIP_GEMM_DEBUG
constexpr bool DBG = true;       // tools/gemm_bench ablations (GemmParams::debug); never compiled into libsvhip.so
#else
constexpr bool DBG = false;
#endif
constexpr int PBM = 256;
constexpr int ROWB = 128;
constexpr int A_TILE = PBM * ROWB;          // 32 KiB
constexpr int PGROUP_M = 4;
// Two tile shapes share the kernel body:
//   BN = 128: 3-stage ring of (32 + 16) KiB, waves 4 (M) x 2 (N), 64 x 64 per wave   (small N, fp32 outputs)
//   BN = 256: 2-stage ring of (32 + 32) KiB, waves 2 (M) x 4 (N), 128 x 64 per wave  (half the LDS-DMA bytes per FLOP)
template <int BN> struct TileCfg {
    static constexpr int B_TILE = BN * ROWB;
    static constexpr int STAGE = A_TILE + B_TILE;
    static constexpr int NSTAGE = BN == 128 ? 3 : 2;
    static constexpr int LDS = NSTAGE * STAGE;                 // 144 KiB / 128 KiB
    static constexpr int WAVES_N = BN / 64, WAVES_M = 8 / WAVES_N;
    static constexpr int MI = PBM / WAVES_M / 32;              // 32-row MFMA tiles per wave along M: 2 / 4
    static constexpr int LOADS = (PBM + BN) / 64;              // global_load_lds per thread per K-step: 6 / 8
};

enum Epi : int { EPI_NONE = 0, EPI_RELU = 1, EPI_GELU = 2, EPI_RELU_TANH = 3, EPI_LRELU03 = 4, EPI_BN_LRELU03 = 5, EPI_LRELU001 = 6 };

__device__ __forceinline__ int swz(int row, int chunk) { return row * ROWB + ((chunk ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ float gelu_fast(float x) {      // erf by Abramowitz-Stegun 7.1.26, |err| <= 1.5e-7
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float e = __expf(-z * z);
    const float erf_abs = fmaf(-poly * t, e, 1.0f);
    return 0.5f * fmaf(fabsf(x), erf_abs, x);
}

template <typename T, int EPI>
__device__ __forceinline__ float act1(float v) {
    if (EPI == EPI_RELU || EPI == EPI_RELU_TANH) return fmaxf(v, 0.0f);
    if (EPI == EPI_GELU) return sizeof(T) == 2 ? gelu_fast(v) : 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    if (EPI == EPI_LRELU03) return v > 0.0f ? v : 0.3f * v;
    if (EPI == EPI_LRELU001) return v > 0.0f ? v : 0.01f * v;
    return v;
}

template <typename T> struct Frag;
template <> struct Frag<float> {
    typedef f32x4 chunk_t;
    static constexpr int EPC = 4, BK = 32;
    static __device__ __forceinline__ void mma(const chunk_t& w, const chunk_t& x, f32x16& c) {
#pragma unroll
        for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(w[j], x[j], c, 0, 0, 0);
    }
};
template <> struct Frag<bf16_t> {
    typedef bf16x8 chunk_t;
    static constexpr int EPC = 8, BK = 64;
    static __device__ __forceinline__ void mma(const chunk_t& w, const chunk_t& x, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, x, c, 0, 0, 0);
    }
};

template <> struct Frag<f16_t> {           // SVHIP_F16 handles (RawNet2): the bf16 kernel with the fp16 MFMA opcode
    typedef f16x8 chunk_t;
    static constexpr int EPC = 8, BK = 64;
    static __device__ __forceinline__ void mma(const chunk_t& w, const chunk_t& x, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, x, c, 0, 0, 0);
    }
};

// CONV: the A operand is the im2col view of a dilated 1-D convolution (per-lane DMA source = frame
// t + (tap - taps/2)*dil of the same utterance, reflect / zero padded; padded chunks read a zero page).
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

// X3 (T = float only): fp32 operands in memory and LDS, each product formed as THREE bf16 MFMAs on hi / lo-split fragments
// (hi.hi + hi.lo + lo.hi, fp32 accumulate): ~2^-17 relative per product instead of 2^-24, at 3/16 of the fp32 MFMA's cycles.
// The activations are split in registers after the fragment read; the packed weights hold (hi bf16 << 16) | lo bf16 per fp32
// slot (api.hip make_conv, SVHIP_F32X3 handles), so their "split" is two byte permutes per pair.
template <typename T, int EPI, bool OUT_F32, int PBN, bool CONV, bool X3 = false>
__global__ __launch_bounds__(512, 2) void gemm_pw_kernel(GemmParams p) {
    static_assert(!X3 || sizeof(T) == 4, "the split path reads fp32 operands");
    typedef Frag<T> FR;
    typedef typename FR::chunk_t chunk_t;
    typedef TileCfg<PBN> TC;
    constexpr int EPC = FR::EPC, BK = FR::BK;
    constexpr int STAGE = TC::STAGE, NSTAGE = TC::NSTAGE, LOADS_PER_STAGE = TC::LOADS, MI = TC::MI;
    constexpr int WROWS = PBM / TC::WAVES_M;                   // rows of the tile owned by one wave: 64 / 128

    extern __shared__ __attribute__((aligned(16))) char smem[];

    // ---- XCD-aware tile mapping --------------------------------------------------------------
    const int ntm = (p.M + PBM - 1) / PBM, ntn = (p.N + PBN - 1) / PBN;
    const int nwg = ntm * ntn;
    int id = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = id & 7;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
    }
    const int grp = id / (PGROUP_M * ntn);
    const int within = id - grp * (PGROUP_M * ntn);
    const int gm = min(PGROUP_M, ntm - grp * PGROUP_M);
    const int tile_m = grp * PGROUP_M + within % gm;
    const int tile_n = within / gm;
    const int m0 = tile_m * PBM, n0 = tile_n * PBN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / TC::WAVES_N, wn = wave % TC::WAVES_N;

    // ---- DMA geometry: wave w fills 8-row groups g = w + 8j (j < 4: A rows, j >= 4: W rows) ----
    const char* src[LOADS_PER_STAGE];
    int dst[LOADS_PER_STAGE];
    int ct[4] = {0, 0, 0, 0}, clc[4] = {0, 0, 0, 0};     // CONV: frame within its utterance, logical chunk
#pragma unroll
    for (int j = 0; j < LOADS_PER_STAGE; ++j) {
        const int g = wave + 8 * j;                                    // 8-row group: 0..31 A, 32.. W
        const bool isA = j < 4;
        const int r = (isA ? g : g - 32) * 8 + (lane >> 3);            // row inside the A / W tile
        const int lc = (lane & 7) ^ ((r >> 1) & 7);                     // logical chunk this lane fetches
        if (isA) {
            const int m = min(m0 + r, p.M - 1);
            // (CONV: the row start only — the chunk is part of the per-K-step offset)
            src[j] = reinterpret_cast<const char*>(p.A) + ((int64_t)m * p.lda + (CONV ? 0 : lc * EPC)) * sizeof(T);
            dst[j] = g * 1024;
            if (CONV) { ct[j] = m - (m / p.T) * p.T; clc[j] = lc; }
        } else {
            const int n = min(n0 + r, p.Wrows - 1);
            src[j] = reinterpret_cast<const char*>(p.W) + ((int64_t)n * p.Kp + lc * EPC) * sizeof(T);
            dst[j] = A_TILE + (g - 32) * 1024;
        }
    }
    // CONV address: a 32-bit offset from the lane's own row, computed branch-free (a select with a 64-bit multiply or a division
    // in one arm compiles to exec-mask branches between the MFMA groups; see gemm_pw2.hip)
    const float rcin = CONV ? 1.0f / (float)p.cin : 0.0f;
    const bool reflect = p.pad_mode == PAD_REFLECT;
    const int ldab = p.lda * (int)sizeof(T), half = p.taps >> 1;
    auto issue = [&](int stage, int kt) {
        char* base = smem + stage * STAGE;
#pragma unroll
        for (int j = 0; j < LOADS_PER_STAGE; ++j) {
            const char* s = src[j] + (int64_t)kt * ROWB;
            if (CONV && j < 4) {
                const int k = kt * BK + clc[j] * EPC;
                const int tap = (int)(((float)k + 0.5f) * rcin);            // k / cin, exact for k < 2^16, cin <= 2^10 (checked on the host)
                const int t0 = ct[j];
                const int tt = t0 + (tap - half) * p.dil;
                const bool inside = (unsigned)tt < (unsigned)p.T;
                const int tr = reflect_idx(tt, p.T);                         // == tt when inside
                const bool ok = (k < p.K) & (inside | reflect);
                int offs = (tr - t0) * ldab + (k - tap * p.cin) * (int)sizeof(T);
                asm volatile("" : "+v"(offs));
                uint64_t q = reinterpret_cast<uint64_t>(src[j]) + (int64_t)offs;
                asm volatile("" : "+v"(q));
                s = reinterpret_cast<const char*>(ok ? q : reinterpret_cast<uint64_t>(p.zero_page));
            }
            __builtin_amdgcn_global_load_lds((gbl_void*)s, (lds_void*)(base + dst[j]), 16, 0, 0);
        }
    };

    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int nk = p.Kp / BK;
    issue(0, 0);
    if (NSTAGE == 3 && nk > 1) issue(1, 1);

    const int fr = lane & 31, fh = lane >> 5;
    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
        // stage kt has landed once all but the newest in-flight K-step of this wave's DMAs are retired
        if (NSTAGE == 3 && kt + 1 < nk && !(DBG && (p.debug & 1))) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (NSTAGE == 3) { if (kt + 2 < nk && !(DBG && (p.debug & 1))) issue(stage >= 1 ? stage - 1 : 2, kt + 2); }   // (stage + 2) % 3
        else { if (kt + 1 < nk && !(DBG && (p.debug & 1))) issue(stage ^ 1, kt + 1); }
        const char* As = smem + stage * STAGE;
        const char* Bs = As + A_TILE;
        if (DBG && (p.debug & 2)) { stage = stage == NSTAGE - 1 ? 0 : stage + 1; continue; }
        if constexpr (X3) {
            // lane (fr, fh) holds chunk 2s + fh of its row for s = 0..3; chunks (2s, 2s+1) x (s, s+1) = 16 k-values = one bf16 MFMA step
            f32x4 xc[4][MI], wc[4][2];
#pragma unroll
            for (int ss = 0; ss < 4; ++ss) {
#pragma unroll
                for (int i = 0; i < MI; ++i) xc[ss][i] = *reinterpret_cast<const f32x4*>(As + swz(wm * WROWS + i * 32 + fr, 2 * ss + fh));
#pragma unroll
                for (int j = 0; j < 2; ++j) wc[ss][j] = *reinterpret_cast<const f32x4*>(Bs + swz(wn * 64 + j * 32 + fr, 2 * ss + fh));
            }
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                bf16x8 xh[MI], xl[MI], wh[2], wl[2];
#pragma unroll
                for (int i = 0; i < MI; ++i) split_hi_lo(xc[2 * pr][i], xc[2 * pr + 1][i], xh[i], xl[i]);
#pragma unroll
                for (int j = 0; j < 2; ++j) {        // the weights were split at load time: word = (hi bf16 << 16) | lo bf16 (two v_perm per two k)
                    u32x4 wa, wb, ph, pl;
                    __builtin_memcpy(&wa, &wc[2 * pr][j], 16);
                    __builtin_memcpy(&wb, &wc[2 * pr + 1][j], 16);
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        ph[e] = __builtin_amdgcn_perm(wa[2 * e + 1], wa[2 * e], 0x07060302u);
                        pl[e] = __builtin_amdgcn_perm(wa[2 * e + 1], wa[2 * e], 0x05040100u);
                        ph[2 + e] = __builtin_amdgcn_perm(wb[2 * e + 1], wb[2 * e], 0x07060302u);
                        pl[2 + e] = __builtin_amdgcn_perm(wb[2 * e + 1], wb[2 * e], 0x05040100u);
                    }
                    __builtin_memcpy(&wh[j], &ph, 16);
                    __builtin_memcpy(&wl[j], &pl, 16);
                }
                // term-major: consecutive MFMAs hit different accumulators (small terms first)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = X3H::mfma32(wl[j], xh[i], acc[i][j]);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = X3H::mfma32(wh[j], xl[i], acc[i][j]);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = X3H::mfma32(wh[j], xh[i], acc[i][j]);
            }
            stage = stage == NSTAGE - 1 ? 0 : stage + 1;
            continue;
        }
        // fragment reads are software-pipelined one k-substep ahead of the MFMAs that consume them
        chunk_t xf[2][MI], wf[2][2];
#pragma unroll
        for (int i = 0; i < MI; ++i) xf[0][i] = *reinterpret_cast<const chunk_t*>(As + swz(wm * WROWS + i * 32 + fr, fh));
#pragma unroll
        for (int j = 0; j < 2; ++j) wf[0][j] = *reinterpret_cast<const chunk_t*>(Bs + swz(wn * 64 + j * 32 + fr, fh));
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int cur = s & 1, nxt = cur ^ 1;
            if (s + 1 < 4 && !(DBG && (p.debug & 16))) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    xf[nxt][i] = *reinterpret_cast<const chunk_t*>(As + swz(wm * WROWS + i * 32 + fr, 2 * (s + 1) + fh));
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    wf[nxt][j] = *reinterpret_cast<const chunk_t*>(Bs + swz(wn * 64 + j * 32 + fr, 2 * (s + 1) + fh));
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) FR::mma(wf[cur][j], xf[cur][i], acc[i][j]);
        }
        stage = stage == NSTAGE - 1 ? 0 : stage + 1;
    }
    __builtin_amdgcn_s_barrier();            // every wave is done reading the ring: reuse it for the output tile
    if (DBG && (p.debug & 4)) { float t = 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i) t += acc[i][0][i] + acc[i][1][3];
        if (t == 123.456f) reinterpret_cast<float*>(p.Y)[tid] = t;
        return; }

    // ---- epilogue: acc[i][j][4g+e] is (m = wm*64 + i*32 + fr, n = wn*64 + j*32 + 8g + 4fh + e) ----
    typedef typename std::conditional<OUT_F32, float, typename std::conditional<sizeof(T) == 2, T, bf16_t>::type>::type OT;
    constexpr int ORB = PBN * (int)sizeof(OT);                    // output tile row bytes: 256 or 512
    constexpr int OCH = OUT_F32 ? 16 : 8;                         // bytes written per lane per group
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int nl = wn * 64 + j * 32 + 8 * g + 4 * fh;     // local column of 4 consecutive channels
            const int n = n0 + nl;
            const bool nok = n < p.N;
            f32x4 b4 = {0.f, 0.f, 0.f, 0.f}, sc4 = {1.f, 1.f, 1.f, 1.f}, sh4 = {0.f, 0.f, 0.f, 0.f};
            if (nok) {
                if (p.bias) b4 = *reinterpret_cast<const f32x4*>(p.bias + n);
                if (p.scale) { sc4 = *reinterpret_cast<const f32x4*>(p.scale + n); sh4 = *reinterpret_cast<const f32x4*>(p.shift + n); }
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int ml = wm * WROWS + i * 32 + fr;
                f32x4 bu = {0.f, 0.f, 0.f, 0.f};
                if (p.bias_utt && nok) {
                    const int m = min(m0 + ml, p.M - 1);
                    bu = *reinterpret_cast<const f32x4*>(p.bias_utt + (int64_t)(m / p.T) * p.ld_bu + n);
                }
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = acc[i][j][4 * g + e] + b4[e] + bu[e];
                    t = act1<T, EPI>(t);
                    t = fmaf(t, sc4[e], sh4[e]);
                    if (EPI == EPI_RELU_TANH) t = tanhf(t);
                    if (EPI == EPI_BN_LRELU03) t = t > 0.0f ? t : 0.3f * t;
                    v[e] = t;
                }
                if (p.R && nok) {                                           // residual (RawNet2 conv2 + shortcut)
                    const int m = min(m0 + ml, p.M - 1);
                    const T* rp = reinterpret_cast<const T*>(p.R) + (int64_t)m * p.ldr + n;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += to_f32<T>(rp[e]);
                }
                if (OUT_F32) {
                    const int c16 = nl >> 2;                                  // 16-byte chunk index
                    f32x4 o = {v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<f32x4*>(smem + ml * ORB + ((c16 ^ (ml & 7)) << 4)) = o;
                } else {
                    const int c8 = nl >> 2;                                   // 8-byte chunk index
                    typedef OT bf16x4 __attribute__((ext_vector_type(4)));
                    bf16x4 o = {static_cast<OT>(v[0]), static_cast<OT>(v[1]), static_cast<OT>(v[2]), static_cast<OT>(v[3])};
                    *reinterpret_cast<bf16x4*>(smem + ml * ORB + ((c8 ^ (ml & 15)) << 3)) = o;
                }
            }
        }
    }
    __syncthreads();
    // ---- whole rows out: 16-byte stores, 16 (bf16) or 32 (fp32) lanes per 256/512-byte row ----------
    constexpr int CPR = ORB / 16;                                             // 16-byte chunks per row
    constexpr int ITER = PBM * CPR / 512;
    (void)OCH;
    char* Yb = reinterpret_cast<char*>(p.Y);
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int idx = it * 512 + tid;
        const int row = idx / CPR, q = idx % CPR;
        u32x4 d;
        if (OUT_F32) {
            d = *reinterpret_cast<const u32x4*>(smem + row * ORB + ((q ^ (row & 7)) << 4));
        } else {
            const int rr = row & 15;
            const u32x4 t = *reinterpret_cast<const u32x4*>(smem + row * ORB + (((2 * q) ^ (rr & 14)) << 3));
            d = (rr & 1) ? u32x4{t[2], t[3], t[0], t[1]} : t;
        }
        const int m = m0 + row;
        const int n = n0 + q * (16 / (int)sizeof(OT));
        if (m < p.M && n < p.N) *reinterpret_cast<u32x4*>(Yb + ((int64_t)m * p.ldy + n) * sizeof(OT)) = d;
    }
}

template <typename T, int EPI, bool OUT_F32, int BN, bool CONV, bool X3>
hipError_t launch_inst3(const GemmParams& p, hipStream_t stream) {
    const int ntm = (p.M + PBM - 1) / PBM, ntn = (p.N + BN - 1) / BN;
    static DeviceOnce attr;
    if (hipError_t e = set_max_dynamic_lds(attr, reinterpret_cast<const void*>(gemm_pw_kernel<T, EPI, OUT_F32, BN, CONV, X3>), TileCfg<BN>::LDS)) return e;
    hipLaunchKernelGGL((gemm_pw_kernel<T, EPI, OUT_F32, BN, CONV, X3>), dim3(ntm * ntn), dim3(512), TileCfg<BN>::LDS, stream, p);
    return hipGetLastError();
}

template <typename T, int EPI, bool OUT_F32, int BN, bool CONV>
hipError_t launch_inst2(const GemmParams& p, hipStream_t stream) {
    if constexpr (std::is_same<T, float>::value) {
        if (p.x3) return launch_inst3<T, EPI, OUT_F32, BN, CONV, true>(p, stream);
    }
    return launch_inst3<T, EPI, OUT_F32, BN, CONV, false>(p, stream);
}

template <typename T, int EPI, bool OUT_F32, int BN>
hipError_t launch_inst(const GemmParams& p, hipStream_t stream) {
    return p.taps > 1 ? launch_inst2<T, EPI, OUT_F32, BN, true>(p, stream) : launch_inst2<T, EPI, OUT_F32, BN, false>(p, stream);
}

template <typename T, bool OUT_F32, int BN>
hipError_t launch_epi(const GemmParams& p, hipStream_t stream) {
    if (p.act1 == ACT_NONE && p.act2 == ACT_NONE) return launch_inst<T, EPI_NONE, OUT_F32, BN>(p, stream);
    if constexpr (!std::is_same<T, f16_t>::value) {      // (fp16: RawNet2's epilogues only)
    if (p.act1 == ACT_RELU && p.act2 == ACT_NONE) return launch_inst<T, EPI_RELU, OUT_F32, BN>(p, stream);
    if (p.act1 == ACT_GELU && p.act2 == ACT_NONE) return launch_inst<T, EPI_GELU, OUT_F32, BN>(p, stream);
    if (p.act1 == ACT_RELU && p.act2 == ACT_TANH) return launch_inst<T, EPI_RELU_TANH, OUT_F32, BN>(p, stream);
    }
    if (p.act1 == ACT_LRELU03 && p.act2 == ACT_NONE) return launch_inst<T, EPI_LRELU03, OUT_F32, BN>(p, stream);
    if (p.act1 == ACT_NONE && p.act2 == ACT_LRELU03) return launch_inst<T, EPI_BN_LRELU03, OUT_F32, BN>(p, stream);
    if (p.act1 == ACT_LRELU001 && p.act2 == ACT_NONE) return launch_inst<T, EPI_LRELU001, OUT_F32, BN>(p, stream);
    return hipErrorInvalidValue;
}

}  // namespace

bool gemm_pw_supported(const GemmParams& p, bool bf16) {
    const int epc = bf16 ? 8 : 4;
    const int bk = bf16 ? 64 : 32;
    if (p.A2 || p.A3) return false;
    if (p.Kp % bk != 0) return false;
    if (p.taps > 1) {        // conv-gather: 16-byte chunks must not straddle taps; padded k / frames read the zero page
        if (!p.zero_page || p.cin % epc != 0 || p.taps * p.cin != p.K || p.T <= 0 || p.M % p.T != 0) return false;
        if (p.pad_mode == PAD_REFLECT && (p.taps / 2) * p.dil >= p.T) return false;
        if (p.Kp >= 65536 || p.cin > 1024) return false;       // the kernel's k / cin is a float multiply
    } else if (p.K != p.Kp) {
        return false;                                        // every K chunk of every row must be real data
    }
    if (p.R && (p.ldr % 4 != 0 || (reinterpret_cast<uintptr_t>(p.R) & 7))) return false;
    const bool out_f32 = !bf16 || p.out_f32;
    if (p.lda % epc != 0) return false;
    if (out_f32) {
        // 16-byte output chunks = 4 floats: a ragged last chunk may spill into row padding, never into the next row
        if (p.ldy % 4 != 0 || p.ldy < ((p.N + 3) & ~3)) return false;
        if (p.N % 4 != 0 && (p.bias || p.scale || p.bias_utt)) return false;      // per-channel vectors are read as float4
    } else if (p.N % 8 != 0 || p.ldy % 8 != 0) {
        return false;
    }
    if ((reinterpret_cast<uintptr_t>(p.A) | reinterpret_cast<uintptr_t>(p.W) | reinterpret_cast<uintptr_t>(p.Y)) & 15) return false;
    if (p.bias && (reinterpret_cast<uintptr_t>(p.bias) & 15)) return false;
    if (p.scale && ((reinterpret_cast<uintptr_t>(p.scale) | reinterpret_cast<uintptr_t>(p.shift)) & 15)) return false;
    if (p.bias_utt && ((reinterpret_cast<uintptr_t>(p.bias_utt) & 15) || p.ld_bu % 4 != 0)) return false;
    return true;
}

hipError_t launch_gemm_pw(const GemmParams& p, bool bf16, hipStream_t stream, bool narrow) {
    if (!gemm_pw_supported(p, bf16) || p.M <= 0 || p.N <= 0 || p.Wrows < p.N) return hipErrorInvalidValue;
    if (bf16 && p.f16) {
        if (p.out_f32) return launch_epi<f16_t, true, 128>(p, stream);
        if (p.N >= 256 && !narrow) return launch_epi<f16_t, false, 256>(p, stream);
        return launch_epi<f16_t, false, 128>(p, stream);
    }
    if (bf16) {
        if (p.out_f32) return launch_epi<bf16_t, true, 128>(p, stream);
        // 256-wide N tiles halve the LDS-DMA bytes per FLOP; the output tile (256 x 256 bf16) still fits the LDS
        if (p.N >= 256 && !narrow && !(DBG && (p.debug & 32))) return launch_epi<bf16_t, false, 256>(p, stream);
        return launch_epi<bf16_t, false, 128>(p, stream);
    }
    return launch_epi<float, true, 128>(p, stream);
}

}  // namespace svhip
