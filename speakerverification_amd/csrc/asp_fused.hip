// asp_fused.hip — attention logits + softmax over time + attentive statistics + BatchNorm in one
// kernel (bf16 path).  Replaces the asp.conv GEMM, its (M x 3C) fp32 logits round trip through HBM
// and the two-pass pooling kernel.
//
// Reference: AttentiveStatisticsPooling.forward, models/ECAPA_TDNN.py:250-259 and asp_bn (:496):
//   attn = conv1x1(att) + b;  w = softmax_t(attn);  mu = sum_t w x;  sd = sqrt(clamp(sum_t w (x-mu)^2, 1e-12))
//   pooled = BN(cat[mu, sd])
//
// One workgroup (4 waves, one per SIMD) per utterance:
//   * att_b (T x 128 bf16, the tanh(BN(ReLU(.))) output of asp.tdnn) is brought into LDS once by
//     LDS-DMA (swizzled on the source address) and is the MFMA A operand (rows = frames);
//   * per pass of 128 channels each wave owns 32 channels x all 13 frame tiles: 104
//     v_mfma_f32_32x32x16_bf16 leave the whole logit column of a channel in ONE lane pair
//     (208 accumulator registers), so max / exp / sum over time are register reductions plus one
//     cross-half shuffle — the logits never exist in memory;
//   * x (the mfa output) is streamed once: each wave LDS-DMAs its own 32-frame x 32-channel slabs into a private ring
//     of 7 (only its own vmcnt orders them) and accumulates sum e*x and sum e*x^2 in fp32;  var = E_w[x^2] - mu^2.
//     The slab stream runs ahead across passes — 6 slabs (12 KB per wave, 48 KB per CU) stay in flight through the
//     logit MFMAs and the softmax of the next 128 channels: with two slabs in flight the kernel sat at 1.6 TB/s
//     (bytes in flight, not bandwidth, was the bound).
#include "common.h"
#include "kernels.h"

namespace svhip {

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

constexpr int AF_TMAX = 416;                       // 13 frame tiles of 32
constexpr int AF_MT = 13;
constexpr int AF_ATT_BYTES = AF_TMAX * 256;        // att tile: 256-byte rows (128 bf16)
constexpr int AF_SLAB = 32 * 64;                   // one wave's x slab: 32 frames x 32 channels bf16
constexpr int AF_NSLOT = 7;                        // slab ring per wave
constexpr int AF_AHEAD = 6;                        // slabs in flight per wave
constexpr int AF_LDS = AF_ATT_BYTES + 4 * AF_NSLOT * AF_SLAB;   // 104 KiB + 56 KiB = all 160 KiB

__global__ __launch_bounds__(256, 1) void asp_fused_kernel(AspFusedParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* att = smem;
    const int b = blockIdx.x;
    const int T = p.T;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    char* slab = smem + AF_ATT_BYTES + wave * AF_NSLOT * AF_SLAB;

    // ---- att_b -> LDS (rows >= T repeat the last frame; they are masked out of the softmax) ---------
    {
        const char* src = reinterpret_cast<const char*>(p.att) + (int64_t)b * T * 256;
        for (int g = wave; g < AF_TMAX / 4; g += 4) {             // one wave-instruction = 4 rows x 16 chunks
            const int r = g * 4 + (lane >> 4);
            const int lc = (lane & 15) ^ (r & 15);
            __builtin_amdgcn_global_load_lds((gbl_void*)(src + (int64_t)min(r, T - 1) * 256 + lc * 16),
                                             (lds_void*)(att + g * 1024), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    const bf16_t* __restrict__ X = reinterpret_cast<const bf16_t*>(p.X) + (int64_t)b * T * p.ldx;
    const int npass = p.C / 128;
    // x slab stream: slab s = (pass s / 13, frame tile s % 13) of this wave's 32 channels -> ring slot s % 7;
    // 32 rows x 64 B = two wave-instructions per slab
    const int nslab = npass * AF_MT;
    int is_pass = 0, is_mt = 0, is_slot = 0;          // next slab to issue
    auto issue_next = [&]() {
        const int cbase = is_pass * 128 + wave * 32;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int row = q * 16 + (lane >> 2);
            const int t = min(is_mt * 32 + row, T - 1);
            __builtin_amdgcn_global_load_lds((gbl_void*)(X + (int64_t)t * p.ldx + cbase + (lane & 3) * 8),
                                             (lds_void*)(slab + is_slot * AF_SLAB + q * 1024), 16, 0, 0);
        }
        if (++is_mt == AF_MT) { is_mt = 0; ++is_pass; }
        if (++is_slot == AF_NSLOT) is_slot = 0;
    };
    for (int i = 0; i < AF_AHEAD && i < nslab; ++i) issue_next();
    int s_idx = 0, rd_slot = 0;                       // next slab to consume
    for (int pass = 0; pass < npass; ++pass) {
        const int c0 = pass * 128 + wave * 32;
        const int c = c0 + fr;

        // ---- logits for 32 channels x all frames: A = att (rows = frames), B = asp.conv weights ------
        bf16x8 wf[8];
        {
            const char* wsrc = reinterpret_cast<const char*>(p.W) + ((int64_t)(c0 + fr) * p.Kp) * 2 + fh * 16;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) wf[kk] = *reinterpret_cast<const bf16x8*>(wsrc + kk * 32);
        }
        f32x16 acc[AF_MT];
#pragma unroll
        for (int mt = 0; mt < AF_MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][r] = 0.0f;
#pragma unroll
        for (int mt = 0; mt < AF_MT; ++mt) {
            const int row = mt * 32 + fr;
            const char* ap = att + row * 256;
            const int sw = row & 15;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(ap + (((2 * kk + fh) ^ sw) << 4));
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, wf[kk], acc[mt], 0, 0, 0);
            }
        }
        // acc[mt][r]: frame t = mt*32 + (r&3) + 8*(r>>2) + 4*fh, channel c (this lane).
        // softmax over t is invariant to the per-channel bias of asp.conv (constant in t), so it is never added; frames >= T
        // (tail of the last tile) are masked to -inf; exp(l - mx) = exp2(l*log2e - mx*log2e): one fma + v_exp.
        float mx = -INFINITY;
#pragma unroll
        for (int mt = 0; mt < AF_MT; ++mt) {
            if ((mt + 1) * 32 > T) {                 // uniform: only tiles that reach past the utterance (the last one at T = 401)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int t = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                    if (t >= T) acc[mt][r] = -INFINITY;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; r += 2) mx = __builtin_fmaxf(mx, __builtin_fmaxf(acc[mt][r], acc[mt][r + 1]));      // v_max3_f32
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        constexpr float L2E = 1.44269504088896340736f;
        const float nmx = -mx * L2E;
        float se = 0.0f;
#pragma unroll
        for (int mt = 0; mt < AF_MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __builtin_amdgcn_exp2f(fmaf(acc[mt][r], L2E, nmx));      // exp2(-inf) = 0 for masked frames
                acc[mt][r] = e;
                se += e;
            }
        // ---- weighted first / second moments of x, slab by slab --------------------------------------------
        float sx = 0.0f, sxx = 0.0f;
#pragma unroll
        for (int mt = 0; mt < AF_MT; ++mt) {
            // slot (s_idx + 6) % 7 == (s_idx - 1) % 7 was consumed in the previous iteration: restage it, then wait for slab
            // s_idx itself (own DMAs only): at most min(6, slabs left) newer slabs may stay in flight
            asm volatile("" ::: "memory");
            if (s_idx + AF_AHEAD < nslab) {
                issue_next();
                asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            } else {
                const int left = nslab - 1 - s_idx;
                if (left >= 5) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
                else if (left == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if (left == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else if (left == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else if (left == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            const char* sp = slab + rd_slot * AF_SLAB + fr * 2;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int tl = (r & 3) + 8 * (r >> 2) + 4 * fh;
                const float xv = static_cast<float>(*reinterpret_cast<const bf16_t*>(sp + tl * 64));
                const float ex = acc[mt][r] * xv;
                sx += ex;
                sxx = fmaf(ex, xv, sxx);
            }
            ++s_idx;
            if (++rd_slot == AF_NSLOT) rd_slot = 0;
        }
        se += __shfl_xor(se, 32, 64);
        sx += __shfl_xor(sx, 32, 64);
        sxx += __shfl_xor(sxx, 32, 64);
        if (fh == 0) {
            const float mean = sx / se;
            const float var = sxx / se - mean * mean;
            const float sd = sqrtf(fmaxf(var, p.eps));
            const int64_t o = (int64_t)b * 2 * p.C;
            if (p.pooled_raw) { p.pooled_raw[o + c] = mean; p.pooled_raw[o + p.C + c] = sd; }
            p.pooled_bn[o + c] = fmaf(mean, p.bn_scale[c], p.bn_shift[c]);
            p.pooled_bn[o + p.C + c] = fmaf(sd, p.bn_scale[p.C + c], p.bn_shift[p.C + c]);
        }
    }
}

}  // namespace

bool asp_fused_supported(int T, int C, int att_channels, int Kp) {
    return T <= AF_TMAX && T >= 1 && C % 128 == 0 && att_channels == 128 && Kp == 128;
}

hipError_t launch_asp_fused(const AspFusedParams& p, int B, hipStream_t stream) {
    if (!asp_fused_supported(p.T, p.C, 128, p.Kp) || B <= 0 || p.ldx % 8 != 0) return hipErrorInvalidValue;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(asp_fused_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, AF_LDS);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(asp_fused_kernel, dim3(B), dim3(256), AF_LDS, stream, p);
    return hipGetLastError();
}

}  // namespace svhip
