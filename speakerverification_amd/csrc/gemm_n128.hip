// gemm_n128.hip — bf16 pointwise GEMM with N = 128 output channels: 128 x 128 tiles, four waves, two workgroups per CU (gfx950).
//
// The layer: asp.tdnn of the attentive statistics pooling (models/ECAPA_TDNN.py:245-250, :68-69): att = tanh(BN(ReLU(W x + ctx[b]))),
// W (128, 3C) over the time-varying columns of cat[x, mean, std], ctx[b] = the per-utterance contribution of the time-constant
// columns (asp_ctx, a small linear layer); M = B T rows, K = 3C = 3072, N = 128.
//
// Why its own kernel (round 4).  N = 128 is half an N tile of the 256 x 256 kernels; on gemm_pw's 256 x 128 tile (one workgroup per CU,
// 3-stage ring) the layer ran at 440 TFLOP/s at B = 256 (0.18 ms, 401 tiles = 1.57 rounds of 256 CUs) and — what hurt more — took 141 us
// at the reference API's own batch size (B = 20: 32 workgroups, each walking 48 K steps alone on its CU: a fifth of a 0.71 ms call).
// Here a tile is 128 frames x 128 channels on four waves (64 x 64 per wave, v_mfma_f32_16x16x32_bf16, weights as the A operand so a lane
// owns 4 consecutive channels of a frame) with 80 KiB of LDS (round 6: three K tiles of X, two of W, 16 KiB each), so TWO workgroups share a CU: one's
// wait for its next K tile (wait, barrier, issue the tiles after next, multiply) runs under the other's
// MFMAs, twice as many workgroups exist for a small batch, and the tail round of a big one is half as long.  Structure borrowed from
// r2_step.hip (the F32X3 Res2Net step), which showed that this shape feeds the matrix pipe at 1.3 PFLOP/s of issue.
// Epilogue: + ctx[row's utterance] -> ReLU -> BN affine -> tanh (1 - 2 / (1 + e^2x) on v_exp / v_rcp: the output is rounded to bf16)
// -> bf16 -> XOR-swizzled LDS image -> whole 256-byte rows in 16-byte stores.
#include "common.h"
#include "kernels.h"
#include "gemm_epi.h"

namespace svhip {

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

constexpr int NT_TILE = 128;                       // frames per tile = channels
constexpr int NT_HT = NT_TILE * 128;               // one operand buffer of one K tile: 128 rows x 64 k bf16 (128 bytes)
constexpr int NT_LDS = 5 * NT_HT;                  // three K tiles of X + two of W = 80 KiB: two workgroups fill a CU's 160 KiB (the 32 KiB output image reuses it)

// (Round 6, measured and dropped: equal row ranges per workgroup — 512 workgroups of 208 rows walked as a pass of 128 and one of 80 instead of 804
//  tiles on 512 slots.  160 us against 148 - 151: a pass costs its 48 K tiles of barriers whatever it streams.)
__global__ __launch_bounds__(256, 2) void gemm_n128_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;        // rows wm * 64 .. + 63, channels wn * 64 .. + 63
    const int r16 = lane & 15, q4 = lane >> 4;
    const int m0 = blockIdx.x * NT_TILE;
    const int mend = p.M;

    // ---- operand DMA addressing: thread -> four (row, 16-byte slot) items of a 128 x 128-byte buffer; the swizzle
    //      (slot ^ (row >> 1 & 7)) goes on the source chunk (the DMA writes LDS lane-linear) ----
    const char* Ab = reinterpret_cast<const char*>(p.A);
    const char* Wb = reinterpret_cast<const char*>(p.W);
    uint32_t xo[4], wo[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int pidx = q * 256 + tid;
        const int row = pidx >> 3, slot = pidx & 7;
        const int c = slot ^ ((row >> 1) & 7);
        const int m = min(m0 + row, mend - 1);
        xo[q] = (uint32_t)m * (uint32_t)(p.lda * 2) + (uint32_t)c * 16u;          // (M * lda * 2 < 2^32: host check)
        wo[q] = (uint32_t)row * (uint32_t)(p.Kp * 2) + (uint32_t)c * 16u;          // weight row = output channel `row`
    }
    // X: a ring of three K tiles (slots 0 .. 2), W: two (slots 3, 4).  X is the HBM stream (read once, 632 MB per launch at B = 256); with one
    // K tile of it in flight per workgroup (round 4/5: plain double buffering) a CU has 32 KiB outstanding and the chip 8 MB — 4.3 TB/s at
    // the latency the stream sees; two tiles in flight double that.  W (768 KB, every workgroup reads all of it) comes from L2.
    auto issue_x = [&](int kt, int slot) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_global_load_lds((gbl_void*)(Ab + xo[q] + kt * 128), (lds_void*)(smem + slot * NT_HT + (q * 256 + wave * 64) * 16), 16, 0, CPOL_NT);      // (one N tile: X is read once)
    };
    auto issue_w = [&](int kt, int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_global_load_lds((gbl_void*)(Wb + wo[q] + kt * 128), (lds_void*)(smem + (3 + buf) * NT_HT + (q * 256 + wave * 64) * 16), 16, 0, 0);
    };
    auto lds_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // acc[i][j][e]: frame m0 + wm*64 + i*16 + r16, channel wn*64 + j*16 + 4*q4 + e
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nkt = p.Kp >> 6;
    issue_x(0, 0);
    issue_w(0, 0);
    if (nkt > 1) issue_x(1, 1);
    const int xrow = (wm * 64 + r16) * 128, wrow = (wn * 64 + r16) * 128;
    const int xkey = ((wm * 64 + r16) >> 1) & 7, wkey = ((wn * 64 + r16) >> 1) & 7;      // (+ 16 i keeps (row >> 1) & 7)
    int xslot = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        // X(kt) and W(kt) have landed; X(kt + 1) — the four youngest requests — stays in flight
        if (kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();                              // ... for every wave; nobody still reads W's other buffer or X's slot of tile kt - 1
        if (kt + 1 < nkt) issue_w(kt + 1, buf ^ 1);
        if (kt + 2 < nkt) issue_x(kt + 2, xslot == 0 ? 2 : xslot - 1);      // (kt + 2) % 3
        const char* xb = smem + xslot * NT_HT + xrow;
        const char* wb = smem + (3 + buf) * NT_HT + wrow;
        xslot = xslot == 2 ? 0 : xslot + 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 wf[4], xf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(wb + j * 2048 + (((ks * 4 + q4) ^ wkey) << 4));
#pragma unroll
            for (int i = 0; i < 4; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(xb + i * 2048 + (((ks * 4 + q4) ^ xkey) << 4));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
        }
    }
    lds_barrier();                                  // every wave is past its last fragment read: the LDS becomes the output image

    // ---- epilogue ----
    constexpr int ORB = 256;                        // bytes per output row (128 bf16)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int nl = wn * 64 + j * 16 + 4 * q4;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(p.scale + nl);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(p.shift + nl);
        f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) b4 = *reinterpret_cast<const f32x4*>(p.bias + nl);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ml = wm * 64 + i * 16 + r16;
            const int m = min(m0 + ml, mend - 1);
            f32x4 bu = {0.f, 0.f, 0.f, 0.f};
            if (p.bias_utt) bu = *reinterpret_cast<const f32x4*>(p.bias_utt + (int64_t)(m / p.T) * p.ld_bu + nl);
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = fmaxf(acc[i][j][e] + b4[e] + bu[e], 0.0f);
                t = fmaf(t, sc[e], sh[e]);
                // tanh(t) = 1 - 2 / (1 + e^(2t)); e^(2t) -> inf gives 1, -> 0 gives -1
                const float ex = __builtin_amdgcn_exp2f(t * 2.8853900817779268f);
                v[e] = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + ex);
            }
            // 8-byte chunk c8 of row ml lives at chunk c8 ^ (ml & 15) (gemm_pw2's output image)
            *reinterpret_cast<uint2*>(smem + ml * ORB + (((nl >> 2) ^ (ml & 15)) << 3)) = make_uint2(bf16_pack2(v[0], v[1]), bf16_pack2(v[2], v[3]));
        }
    }
    lds_barrier();
    char* Yb = reinterpret_cast<char*>(p.Y);
#pragma unroll
    for (int it = 0; it < 8; ++it) {                // 128 rows x 16 chunks of 16 bytes / 256 threads
        const int idx = it * 256 + tid;
        const int row = idx >> 4, q = idx & 15;
        const int rr = row & 15;
        const u32x4 t = *reinterpret_cast<const u32x4*>(smem + row * ORB + (((2 * q) ^ (rr & 14)) << 3));
        const u32x4 d = (rr & 1) ? u32x4{t[2], t[3], t[0], t[1]} : t;
        if (m0 + row < mend) *reinterpret_cast<u32x4*>(Yb + ((int64_t)(m0 + row) * p.ldy + q * 8) * 2) = d;
    }
}

}  // namespace

// bf16 in / out, pointwise, N = 128, K = Kp a multiple of 64, ReLU -> BN affine -> tanh, optional per-channel and per-utterance bias
bool gemm_n128_supported(const GemmParams& p) {
    if (p.f16 || p.x3 || p.out_f32 || p.taps != 1 || p.A2 || p.A3 || p.R || p.colsum) return false;
    if (p.N != 128 || p.Wrows < 128 || p.K != p.Kp || p.Kp % 64 != 0 || p.Kp < 64) return false;
    if (p.act1 != ACT_RELU || p.act2 != ACT_TANH || !p.scale || !p.shift) return false;
    if (p.lda < p.K || p.lda % 8 != 0 || p.ldy < 128 || p.ldy % 8 != 0 || p.M <= 0) return false;
    if (p.bias_utt && (p.T <= 0 || p.ld_bu % 4 != 0 || (reinterpret_cast<uintptr_t>(p.bias_utt) & 15))) return false;
    if ((int64_t)p.M * p.lda * 2 >= ((int64_t)1 << 32) || (int64_t)128 * p.Kp * 2 >= ((int64_t)1 << 31)) return false;
    if ((reinterpret_cast<uintptr_t>(p.A) | reinterpret_cast<uintptr_t>(p.W) | reinterpret_cast<uintptr_t>(p.Y) | reinterpret_cast<uintptr_t>(p.scale) |
         reinterpret_cast<uintptr_t>(p.shift) | reinterpret_cast<uintptr_t>(p.bias)) & 15) return false;
    return true;
}

hipError_t launch_gemm_n128(const GemmParams& p, hipStream_t stream) {
    if (!gemm_n128_supported(p)) return hipErrorInvalidValue;
    static DeviceOnce attr;
    if (hipError_t e = set_max_dynamic_lds(attr, reinterpret_cast<const void*>(gemm_n128_kernel), NT_LDS)) return e;
    hipLaunchKernelGGL(gemm_n128_kernel, dim3((p.M + NT_TILE - 1) / NT_TILE), dim3(256), NT_LDS, stream, p);
    return hipGetLastError();
}

}  // namespace svhip
