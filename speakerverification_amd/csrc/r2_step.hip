// r2_step.hip — one step of a Res2Net chain on SVHIP_F32X3 handles, 128 x 128 tiles, two workgroups per CU (gfx950).
//
// Same contract as gemm_pw3's R2 form (models/ECAPA_TDNN.py:118-129): y_j = BN(ReLU(conv_k3_dilated(U_j))) with U_j = c_j + y_{j-1}
// in the S32 split layout (per row, per 32 channels: 32 hi bf16 | 32 lo bf16), products as bf16 MFMA triples (hi.hi + hi.lo + lo.hi);
// y_j goes to the chain output in S32, U_{j+1} = y_j + c_{j+1} (c from the fp32 tdnn1 output) to the next step's input.
//
// Why a second kernel.  The R2 form borrows gemm_pw3's 256 x 256 tile with N = 128 channels: two of its four wave columns hold no
// channels, so only two of a CU's four SIMDs do useful matrix work, and its epilogue needs the whole 128 KiB operand ring for the
// c tile, so K loop (matrix-bound) and epilogue (HBM-bound: c in, y and U out, 384 KB per tile) alternate on every CU at once
// (88 us per step; gemm_pw3.hip's header).  Here a workgroup is four waves on a 128-frame x 128-channel tile — every SIMD holds a
// useful wave — with 64 KiB of LDS (two K-tile buffers of X and W, 16 KiB each; the same 64 KiB then stage the c / U / y images),
// so TWO workgroups share a CU and one's epilogue runs under the other's K loop.
#include "common.h"
#include "kernels.h"
#include "gemm_epi.h"

namespace svhip {

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

constexpr int RS_TILE = 128;                       // frames per tile = channels per tile
constexpr int RS_HT = RS_TILE * 128;               // one operand half-buffer: 128 rows x one 32-k block (128 bytes)
constexpr int RS_LDS = 4 * RS_HT;                  // {X, W} x two K tiles = 64 KiB = one 128-row image of 512-byte rows

// MODE 0: the Res2Net step above.  MODE 1 / 2 (round 4, late): RawNet2's 128 -> 128 convolutions on F32X3 handles (models/RawNet_baseline.py
// :224-227) — zero padding (rows outside the utterance read a zero page), no bias; 1 = conv1: BN -> LeakyReLU(0.3), output in the S32 layout
// (conv2's operand); 2 = conv2: no activation, fp32 output = acc + R (the identity shortcut x, fp32) through the same 64 KiB row image;
// 3 = 2 followed by max_pool1d(3) (:228-229): tiles of 126 frames inside ONE utterance (42 pooled frames; the last two MFMA rows are idle),
// the pooled rows leave from the image — the un-pooled conv2 output (1.4 GB for layer1 at B = 256) is never written or read back.
template <int MODE>
__global__ __launch_bounds__(256, 2) void r2_step_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;        // rows wm * 64 .. + 63, channels wn * 64 .. + 63
    const int r16 = lane & 15, q4 = lane >> 4;
    const int n0 = MODE == 0 ? 0 : (int)blockIdx.y * RS_TILE;      // (modes 1 - 3: N may be several tiles of 128 channels)
    // mode 3: workgroup = (utterance bu, tile of 126 frames starting at t0): its rows never cross an utterance
    constexpr int PT = 126;
    const int tiles_u = MODE == 3 ? (3 * (p.T / 3) + PT - 1) / PT : 1;
    const int bu = MODE == 3 ? (int)blockIdx.x / tiles_u : 0;
    const int t0 = MODE == 3 ? ((int)blockIdx.x - bu * tiles_u) * PT : 0;
    const int m0 = MODE == 3 ? bu * p.T + t0 : (int)blockIdx.x * RS_TILE;

    // ---- operand DMA addressing: thread -> four (row, 16-byte slot) items of a 128 x 128-byte half-buffer; the swizzle
    //      (slot ^ (row >> 1 & 7)) goes on the source chunk ----
    const char* Ab = reinterpret_cast<const char*>(p.A);
    const char* Wb = reinterpret_cast<const char*>(p.W);
    int xm[4], xt[4];
    uint32_t xc[4], wo[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int pidx = q * 256 + tid;
        const int row = pidx >> 3, slot = pidx & 7;
        const int c = slot ^ ((row >> 1) & 7);
        const int m = MODE == 3 ? bu * p.T + min(t0 + row, p.T - 1) : min(m0 + row, p.M - 1);
        xm[q] = m;
        xt[q] = MODE == 3 ? (t0 + row < p.T ? t0 + row : -(1 << 20)) : m % p.T;      // (mode 3: rows past the utterance read the zero page)
        xc[q] = (uint32_t)c * 16u;
        wo[q] = (uint32_t)(n0 + row) * (uint32_t)p.Kp * 4u + (uint32_t)c * 16u;     // weight row = output channel n0 + row (< 2^32 bytes: host check)
    }
    auto issue = [&](int kt, int buf) {
        // 32-k blocks per tap: four (cin = 128) in the Res2Net step; cin / 32 in modes 1 / 2 (taps = 1 or 3, dilation 1)
        int tap, kin;
        if (MODE == 0) { tap = kt >> 2; kin = kt & 3; }
        else { const int ktpt = p.cin >> 5; tap = kt / ktpt; kin = kt - tap * ktpt; }
        const int shift = MODE == 0 ? (tap - 1) * p.dil : tap - (p.taps >> 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const char* s;
            if (MODE == 0) {
                const int tt = reflect_idx(xt[q] + shift, p.T);
                s = Ab + ((int64_t)(xm[q] + tt - xt[q]) * p.lda * 4 + kin * 128 + xc[q]);
            } else {
                const bool in = (unsigned)(xt[q] + shift) < (unsigned)p.T;
                s = in ? Ab + ((int64_t)(xm[q] + shift) * p.lda * 4 + kin * 128 + xc[q]) : reinterpret_cast<const char*>(p.zero_page) + xc[q];
            }
            __builtin_amdgcn_global_load_lds((gbl_void*)s, (lds_void*)(smem + (buf * 2) * RS_HT + (q * 256 + wave * 64) * 16), 16, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const char* s = Wb + wo[q] + kt * 128;
            __builtin_amdgcn_global_load_lds((gbl_void*)s, (lds_void*)(smem + (buf * 2 + 1) * RS_HT + (q * 256 + wave * 64) * 16), 16, 0, 0);
        }
    };
    auto lds_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // acc[i][j][e]: frame m0 + wm*64 + i*16 + r16, channel wn*64 + j*16 + 4*q4 + e; starts at the bias
    f32x4 acc[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
        if (MODE == 0 || p.bias) b4 = *reinterpret_cast<const f32x4*>(p.bias + wn * 64 + j * 16 + 4 * q4);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][j] = b4;
    }
    issue(0, 0);
    const int nkt = p.Kp >> 5;                      // 12
    const int xrow = (wm * 64 + r16) * 128, wrow = (wn * 64 + r16) * 128;
    const int xkey = ((wm * 64 + r16) >> 1) & 7, wkey = ((wn * 64 + r16) >> 1) & 7;      // (+ 16 i keeps (row >> 1) & 7)
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();                              // K tile kt has landed for every wave; nobody still reads the other buffer
        if (kt + 1 < nkt) issue(kt + 1, buf ^ 1);
        const char* xb = smem + (buf * 2) * RS_HT + xrow;
        const char* wb = smem + (buf * 2 + 1) * RS_HT + wrow;
        bf16x8 wh[4], wl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            wh[j] = *reinterpret_cast<const bf16x8*>(wb + j * 2048 + ((q4 ^ wkey) << 4));
            wl[j] = *reinterpret_cast<const bf16x8*>(wb + j * 2048 + (((4 + q4) ^ wkey) << 4));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bf16x8 xh = *reinterpret_cast<const bf16x8*>(xb + i * 2048 + ((q4 ^ xkey) << 4));
            const bf16x8 xl = *reinterpret_cast<const bf16x8*>(xb + i * 2048 + (((4 + q4) ^ xkey) << 4));
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = X3H::mfma16(wh[j], xl, acc[i][j]);      // small terms first
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = X3H::mfma16(wl[j], xh, acc[i][j]);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = X3H::mfma16(wh[j], xh, acc[i][j]);
        }
    }
    lds_barrier();                                  // every wave is past its last fragment read: the 64 KiB become the row image

    // ---- epilogue: y = BN(ReLU(acc)); the image (128 rows x 512 bytes, 16-byte chunk k of a 128-byte block at k ^ (row & 7)) holds
    //      first c (fp32), turned in place into U = y + c in S32 (a block is read and written by the four q4 lanes of ONE wave, whose
    //      LDS operations execute in order), then y; each leaves through all four waves as whole rows ----
    const float* Cn = reinterpret_cast<const float*>(p.R);
    char* Ub = reinterpret_cast<char*>(p.Y2);
    auto copy_out = [&](char* dst, int64_t ld_bytes) {
#pragma unroll 4
        for (int q = 0; q < 16; ++q) {
            const int pidx = q * 256 + tid;
            const int row = pidx >> 5, ch = pidx & 31;
            const u32x4 v = *reinterpret_cast<const u32x4*>(smem + row * 512 + (ch & ~7) * 16 + (((ch & 7) ^ (row & 7)) << 4));
            if (m0 + row < p.M) *reinterpret_cast<u32x4*>(dst + (int64_t)(m0 + row) * ld_bytes + ch * 16) = v;
        }
    };
    auto put_s32 = [&](const f32x4& v, int ml, int nl) {
        uint32_t hd[2], ld[2];
#pragma unroll
        for (int d = 0; d < 2; ++d) x3_split2(v[2 * d], v[2 * d + 1], hd[d], ld[d]);
        const int kh = (nl & 31) >> 3;
        char* blk = smem + ml * 512 + (nl >> 5) * 128 + (nl & 4) * 2;
        *reinterpret_cast<uint2*>(blk + ((kh ^ (ml & 7)) << 4)) = make_uint2(hd[0], hd[1]);
        *reinterpret_cast<uint2*>(blk + (((kh + 4) ^ (ml & 7)) << 4)) = make_uint2(ld[0], ld[1]);
    };
    if (MODE == 1) {                                // y = lrelu(BN(acc)) in S32 -> Y
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nl = wn * 64 + j * 16 + 4 * q4;
            const f32x4 sc = *reinterpret_cast<const f32x4*>(p.scale + n0 + nl);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(p.shift + n0 + nl);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float t = fmaf(acc[i][j][e], sc[e], sh[e]); v[e] = fmaxf(t, 0.3f * t); }
                put_s32(v, wm * 64 + i * 16 + r16, nl);
            }
        }
        lds_barrier();
        copy_out(reinterpret_cast<char*>(p.Y) + (int64_t)n0 * 4, (int64_t)p.ldy * 4);      // (128 channels = four S32 blocks = 512 bytes, like 128 floats)
        return;
    }
    if (MODE == 2 || MODE == 3) {                   // out = acc (+ R), fp32, through the row image (mode 3: pooled on the way out)
        if (Cn) {
#pragma unroll 4
            for (int q = 0; q < 16; ++q) {
                const int pidx = q * 256 + tid;
                const int row = pidx >> 5, pos = pidx & 31;
                const int cs = (pos & ~7) | ((pos ^ row) & 7);
                const int m = MODE == 3 ? bu * p.T + min(t0 + row, p.T - 1) : min(m0 + row, p.M - 1);
                __builtin_amdgcn_global_load_lds((gbl_void*)(Cn + (int64_t)m * p.ldr + n0 + (cs << 2)), (lds_void*)(smem + (q * 256 + wave * 64) * 16), 16, 0, CPOL_NT);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_barrier();
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ml = wm * 64 + i * 16 + r16, nl = wn * 64 + j * 16 + 4 * q4;
                f32x4* slot = reinterpret_cast<f32x4*>(smem + ml * 512 + (nl >> 5) * 128 + ((((nl & 31) >> 2) ^ (ml & 7)) << 4));
                *slot = Cn ? acc[i][j] + *slot : acc[i][j];
            }
        lds_barrier();
        if (MODE == 3) {
            // pooled frame pr0 + prow of the utterance = max over image rows 3 prow .. + 2; 42 rows x 32 chunks of 16 bytes
            const int Tp = p.T / 3, pr0 = t0 / 3;
            float* yb = reinterpret_cast<float*>(p.Y) + n0;
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const int pidx = q * 256 + tid;
                const int prow = pidx >> 5, ch = pidx & 31;
                if (prow < PT / 3 && pr0 + prow < Tp) {
                    f32x4 v[3];
#pragma unroll
                    for (int u = 0; u < 3; ++u) {
                        const int row = 3 * prow + u;
                        v[u] = *reinterpret_cast<const f32x4*>(smem + row * 512 + (ch & ~7) * 16 + (((ch & 7) ^ (row & 7)) << 4));
                    }
                    f32x4 m;
#pragma unroll
                    for (int e = 0; e < 4; ++e) m[e] = fmaxf(fmaxf(v[0][e], v[1][e]), v[2][e]);
                    *reinterpret_cast<f32x4*>(yb + ((int64_t)bu * Tp + pr0 + prow) * p.ldy + ch * 4) = m;
                }
            }
            return;
        }
        copy_out(reinterpret_cast<char*>(p.Y) + (int64_t)n0 * 4, (int64_t)p.ldy * 4);
        return;
    }
    if (Cn) {
#pragma unroll 4
        for (int q = 0; q < 16; ++q) {
            const int pidx = q * 256 + tid;          // 16-byte chunk position of the 128 x 32-chunk image
            const int row = pidx >> 5, pos = pidx & 31;
            const int cs = (pos & ~7) | ((pos ^ row) & 7);
            const int m = min(m0 + row, p.M - 1);
            __builtin_amdgcn_global_load_lds((gbl_void*)(Cn + (int64_t)m * p.ldr + (cs << 2)), (lds_void*)(smem + (q * 256 + wave * 64) * 16), 16, 0, CPOL_NT);      // (the next chunk of the tdnn1 output: read once)
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
    }
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
        f32x4 sc[2], sh[2];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int nl = wn * 64 + (2 * jp + jj) * 16 + 4 * q4;
            sc[jj] = *reinterpret_cast<const f32x4*>(p.scale + nl);
            sh[jj] = *reinterpret_cast<const f32x4*>(p.shift + nl);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ml = wm * 64 + i * 16 + r16;
            f32x4 cn[2];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int j = 2 * jp + jj;
                const int nl = wn * 64 + j * 16 + 4 * q4;
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][e] = fmaf(fmaxf(acc[i][j][e], 0.0f), sc[jj][e], sh[jj][e]);
                if (Cn) cn[jj] = *reinterpret_cast<const f32x4*>(smem + ml * 512 + (nl >> 5) * 128 + ((((nl & 31) >> 2) ^ (ml & 7)) << 4));
            }
            if (Cn) {
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) put_s32(acc[i][2 * jp + jj] + cn[jj], ml, wn * 64 + (2 * jp + jj) * 16 + 4 * q4);
            }
        }
    }
    if (Cn) {
        lds_barrier();                              // the U image is complete
        copy_out(Ub, (int64_t)p.lda2 * 4);
        lds_barrier();                              // ... and has been read
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) put_s32(acc[i][j], wm * 64 + i * 16 + r16, wn * 64 + j * 16 + 4 * q4);
    lds_barrier();
    copy_out(reinterpret_cast<char*>(p.Y), (int64_t)p.ldy * 4);
}

}  // namespace

// cin = N = 128, K = Kp = 384, reflect padding, ReLU -> BN; the same GemmParams as gemm_pw3's R2 form
bool r2_step_supported(const GemmParams& p) {
    if (p.x3 != 2 || p.taps != 3 || p.A2 || p.A3 || p.bias_utt || p.colsum || p.out_f32) return false;
    if (p.cin != 128 || p.N != 128 || p.K != 384 || p.Kp != 384 || p.Wrows < 128) return false;
    if (p.act1 != ACT_RELU || p.act2 != ACT_NONE || p.pad_mode != PAD_REFLECT) return false;
    if (!p.bias || !p.scale || !p.shift || !p.Y || !p.A || !p.W) return false;
    if ((p.R == nullptr) != (p.Y2 == nullptr)) return false;
    if (p.lda < 128 || p.lda % 32 != 0 || p.ldy % 32 != 0 || (p.Y2 && (p.lda2 % 32 != 0 || p.ldr % 4 != 0))) return false;
    if (p.T <= 2 * p.dil || p.dil < 1 || p.dil > 8 || p.M <= 0 || p.M % p.T != 0) return false;
    if ((reinterpret_cast<uintptr_t>(p.A) | reinterpret_cast<uintptr_t>(p.W) | reinterpret_cast<uintptr_t>(p.Y) | reinterpret_cast<uintptr_t>(p.Y2) |
         reinterpret_cast<uintptr_t>(p.R) | reinterpret_cast<uintptr_t>(p.bias) | reinterpret_cast<uintptr_t>(p.scale) | reinterpret_cast<uintptr_t>(p.shift)) & 15) return false;
    return true;
}

hipError_t launch_r2_step(const GemmParams& p, hipStream_t stream) {
    if (!r2_step_supported(p)) return hipErrorInvalidValue;
    static DeviceOnce attr;
    if (hipError_t e = set_max_dynamic_lds(attr, reinterpret_cast<const void*>(r2_step_kernel<0>), RS_LDS)) return e;
    hipLaunchKernelGGL(r2_step_kernel<0>, dim3((p.M + RS_TILE - 1) / RS_TILE), dim3(256), RS_LDS, stream, p);
    return hipGetLastError();
}

// RawNet2's convolutions on F32X3 handles (k = 3 with zero padding inside each utterance, or k = 1: the projection shortcuts; cin a multiple of
// 32, N of 128): A (M, lda) and W (N, taps cin) in the S32 layout, zero_page >= 128 bytes of zeros, no bias.  mode 1: Y = lrelu0.3(BN(conv)) in
// S32 (row stride ldy elements); mode 2: Y = conv (+ R) in fp32.  Grid (M tiles, N tiles of 128 channels).
bool rn_step_supported(const GemmParams& p, int mode) {
    if (mode == 3) { if (p.taps != 3 || p.T < 3) return false; mode = 2; }          // (pooled conv2: the contract of mode 2; Y holds (M / T) * (T / 3) rows)
    if (!(mode == 1 || mode == 2) || p.x3 != 2 || !(p.taps == 3 || p.taps == 1) || p.A2 || p.A3 || p.bias_utt || p.colsum || p.bias) return false;
    if (p.cin < 32 || p.cin % 32 != 0 || p.N < 128 || p.N % 128 != 0 || p.K != p.taps * p.cin || p.Kp != p.K || p.Wrows < p.N || p.pad_mode != PAD_ZERO || !p.zero_page) return false;
    if ((int64_t)p.Wrows * p.Kp * 4 >= ((int64_t)1 << 32) || p.N / 128 > 65535) return false;
    if (mode == 1 && (!p.scale || !p.shift || p.out_f32 || p.R || p.ldy % 32 != 0)) return false;
    if (mode == 2 && (!p.out_f32 || p.ldy % 4 != 0 || (p.R && p.ldr % 4 != 0))) return false;
    if (!p.Y || !p.A || !p.W || p.lda < p.cin || p.lda % 32 != 0 || p.ldy < p.N || (p.R && p.ldr < p.N)) return false;
    if (p.T < 2 || p.dil != 1 || p.M <= 0 || p.M % p.T != 0) return false;
    if ((reinterpret_cast<uintptr_t>(p.A) | reinterpret_cast<uintptr_t>(p.W) | reinterpret_cast<uintptr_t>(p.Y) | reinterpret_cast<uintptr_t>(p.R) |
         reinterpret_cast<uintptr_t>(p.zero_page) | reinterpret_cast<uintptr_t>(p.scale) | reinterpret_cast<uintptr_t>(p.shift)) & 15) return false;
    return true;
}

hipError_t launch_rn_step(const GemmParams& p, int mode, hipStream_t stream) {
    if (!rn_step_supported(p, mode)) return hipErrorInvalidValue;
    if (mode == 3) {
        const int tiles_u = (3 * (p.T / 3) + 125) / 126;
        const int64_t gx = (int64_t)(p.M / p.T) * tiles_u;
        if (gx >= ((int64_t)1 << 31)) return hipErrorInvalidValue;
        static DeviceOnce attr3;
        if (hipError_t e = set_max_dynamic_lds(attr3, reinterpret_cast<const void*>(r2_step_kernel<3>), RS_LDS)) return e;
        hipLaunchKernelGGL(r2_step_kernel<3>, dim3((unsigned)gx, p.N / RS_TILE), dim3(256), RS_LDS, stream, p);
        return hipGetLastError();
    }
    const dim3 grid((p.M + RS_TILE - 1) / RS_TILE, p.N / RS_TILE);
    if (mode == 1) {
        static DeviceOnce attr1;
        if (hipError_t e = set_max_dynamic_lds(attr1, reinterpret_cast<const void*>(r2_step_kernel<1>), RS_LDS)) return e;
        hipLaunchKernelGGL(r2_step_kernel<1>, grid, dim3(256), RS_LDS, stream, p);
    } else {
        static DeviceOnce attr2;
        if (hipError_t e = set_max_dynamic_lds(attr2, reinterpret_cast<const void*>(r2_step_kernel<2>), RS_LDS)) return e;
        hipLaunchKernelGGL(r2_step_kernel<2>, grid, dim3(256), RS_LDS, stream, p);
    }
    return hipGetLastError();
}

}  // namespace svhip
