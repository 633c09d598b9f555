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

#ifdef SVHIP_GEMM_DEBUG
constexpr bool AFDBG = true;
#else
constexpr bool AFDBG = false;
#endif
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
                                             (lds_void*)(att + g * 1024), 16, 0, CPOL_NT);
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
    // lane part of the source address (row inside the tile, 16-byte piece) as 32-bit byte offsets for tiles that lie inside
    // the utterance; a tile that reaches past frame T-1 (the last one at T = 401, several for short utterances) clamps its
    // rows to the last frame and computes them on the spot
    uint32_t xoff[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) xoff[q] = ((uint32_t)(q * 16 + (lane >> 2)) * (uint32_t)p.ldx + (uint32_t)(lane & 3) * 8u) * 2u;
    auto issue_next = [&]() {
        const bool tail = (is_mt + 1) * 32 > T;                                                         // uniform
        const char* gb = reinterpret_cast<const char*>(X) + ((int64_t)(tail ? 0 : is_mt * 32) * p.ldx + is_pass * 128 + wave * 32) * 2;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            uint32_t o = xoff[q];
            if (tail) o = ((uint32_t)min(is_mt * 32 + q * 16 + (lane >> 2), T - 1) * (uint32_t)p.ldx + (uint32_t)(lane & 3) * 8u) * 2u;
            asm volatile("" : "+v"(o));
            __builtin_amdgcn_global_load_lds((gbl_void*)(gb + o), (lds_void*)(slab + is_slot * AF_SLAB + q * 1024), 16, 0, CPOL_NT);      // (x is read once: non-temporal)
        }
        if (++is_mt == AF_MT) { is_mt = 0; ++is_pass; }
        if (++is_slot == AF_NSLOT) is_slot = 0;
    };
    for (int i = 0; i < AF_AHEAD && i < nslab; ++i) issue_next();
    int s_idx = 0, rd_slot = 0;                       // next slab to consume
    // transposed slab read: a lane needs x[t0 .. t0+3][its channel] for t0 = 8j + 4fh: a 4-row x 16-channel block per 16-lane
    // group, column-major -> one ds_read_b64_tr_b16 (lane 4q+p supplies row q, channels 4p..4p+3) instead of four 2-byte reads.
    // Inline asm: through the builtin the compiler orders the read behind EVERY pending LDS-DMA with vmcnt(0), draining the ring.
    // The reads of slab s+1 are issued at the end of tile s and waited for (lgkmcnt) at the start of tile s+1, across passes too.
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    u32x2 xr0, xr1, xr2, xr3;
    const uint32_t sp0 = (uint32_t)(uintptr_t)(slab + (4 * fh + ((lane >> 2) & 3)) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8);
    auto read_slab = [&](int slot) {
        const uint32_t sp = sp0 + slot * AF_SLAB;
        asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %4 offset:512\n\t"
                     "ds_read_b64_tr_b16 %2, %4 offset:1024\n\tds_read_b64_tr_b16 %3, %4 offset:1536"
                     : "=&v"(xr0), "=&v"(xr1), "=&v"(xr2), "=&v"(xr3) : "v"(sp) : "memory");
    };
    if (nslab > AF_AHEAD) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");      // slab 0 landed
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    read_slab(0);
    bf16x8 wnext[8];
    auto load_w = [&](int pass_) {
        const char* wsrc = reinterpret_cast<const char*>(p.W) + ((int64_t)(pass_ * 128 + wave * 32 + fr) * p.Kp) * 2 + fh * 16;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) wnext[kk] = *reinterpret_cast<const bf16x8*>(wsrc + kk * 32);
    };
    load_w(0);
    unsigned long long tph[5] = {0, 0, 0, 0, 0};
#define AF_STAMP(v) unsigned long long v = 0; if (AFDBG && p.dbg) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); v = __builtin_readcyclecounter(); }
    for (int pass = 0; pass < npass; ++pass) {
        const int c0 = pass * 128 + wave * 32;
        const int c = c0 + fr;
        AF_STAMP(t0)

        // ---- logits for 32 channels x all frames: A = att (rows = frames), B = asp.conv weights (prefetched during the
        //      previous pass's moments loop) ------
        bf16x8 wf[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) wf[kk] = wnext[kk];
        f32x16 acc[AF_MT];
#pragma unroll
        for (int mt = 0; mt < AF_MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][r] = 0.0f;
        if (AFDBG && p.dbg) { float zz = 0.f;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) zz += static_cast<float>(wf[kk][0]);
            asm volatile("" :: "v"(zz)); }
        AF_STAMP(t1)
        // k-step outer, frame tile inner: 13 independent accumulators between two MFMAs on the same one (a k-inner order
        // chains 8 dependent MFMAs per tile), and the A fragments of step kk+1 are in flight under the MFMAs of step kk
        {
            const char* ap = att + fr * 256;
            const int sw = fr & 15;                              // (mt*32 + fr) & 15
            bf16x8 af[2][AF_MT];
#pragma unroll
            for (int mt = 0; mt < AF_MT; ++mt) af[0][mt] = *reinterpret_cast<const bf16x8*>(ap + mt * 8192 + ((fh ^ sw) << 4));
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                if (kk < 7) {
#pragma unroll
                    for (int mt = 0; mt < AF_MT; ++mt)
                        af[(kk + 1) & 1][mt] = *reinterpret_cast<const bf16x8*>(ap + mt * 8192 + (((2 * (kk + 1) + fh) ^ sw) << 4));
                }
#pragma unroll
                for (int mt = 0; mt < AF_MT; ++mt)
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk & 1][mt], wf[kk], acc[mt], 0, 0, 0);
            }
        }
        if (AFDBG && p.dbg) { float zz = 0.f;
#pragma unroll
            for (int mt = 0; mt < AF_MT; ++mt) zz += acc[mt][0];
            asm volatile("" :: "v"(zz)); }
        AF_STAMP(t2)
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
        float se = 0.0f;                                  // e = exp2(...) is taken tile by tile in the moments loop below (one v_exp per logit)
        if (AFDBG && p.dbg) asm volatile("" :: "v"(nmx));
        AF_STAMP(t3)
        // ---- weighted first / second moments of x, slab by slab --------------------------------------------
        if (pass + 1 < npass) load_w(pass + 1);          // lands under the moments loop
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 se2 = {0.f, 0.f}, sx2 = {0.f, 0.f}, sxx2 = {0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < AF_MT; ++mt) {
            // slot (s_idx + 6) % 7 == (s_idx - 1) % 7 was consumed in the previous iteration: restage it; then make sure slab
            // s_idx + 1 has landed (own DMAs only; 5 newer slabs stay in flight) so that its LDS reads can go out after this tile
            asm volatile("" ::: "memory");
            if (s_idx + AF_AHEAD < nslab) {
                issue_next();
                asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the last six slabs of the stream
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xr0), "+v"(xr1), "+v"(xr2), "+v"(xr3) :: "memory");
            const u32x2 xr[4] = {xr0, xr1, xr2, xr3};
            // two frames at a time on the packed fp32 pipe (v_pk_fma / v_pk_mul / v_pk_add); only the two v_exp are scalar
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const uint32_t w = xr[j][h2];
                    const f32x2 xv = {__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
                    const f32x2 lg = {acc[mt][4 * j + 2 * h2], acc[mt][4 * j + 2 * h2 + 1]};
                    const f32x2 z = lg * f32x2{L2E, L2E} + f32x2{nmx, nmx};
                    const f32x2 e = {__builtin_amdgcn_exp2f(z.x), __builtin_amdgcn_exp2f(z.y)};      // exp2(-inf) = 0 for masked frames
                    se2 += e;
                    const f32x2 ex = e * xv;
                    sx2 += ex;
                    sxx2 = ex * xv + sxx2;
                }
            }
            ++s_idx;
            if (++rd_slot == AF_NSLOT) rd_slot = 0;
            asm volatile("" :: "v"(sx2), "v"(sxx2) : "memory");         // the tile's use of xr ends here
            if (s_idx < nslab) read_slab(rd_slot);
        }
        float sx = sx2.x + sx2.y, sxx = sxx2.x + sxx2.y;
        se = se2.x + se2.y;
        if (AFDBG && p.dbg) asm volatile("" :: "v"(sx), "v"(sxx));
        AF_STAMP(t4)
        if (AFDBG && p.dbg) { tph[0] += t1 - t0; tph[1] += t2 - t1; tph[2] += t3 - t2; tph[3] += t4 - t3; }
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
    if (AFDBG && p.dbg && tid == 0) { for (int i = 0; i < 4; ++i) p.dbg[(int64_t)b * 4 + i] = tph[i]; }
#undef AF_STAMP
}

}  // namespace

bool asp_fused_supported(int T, int C, int att_channels, int Kp) {
    return T <= AF_TMAX && T >= 1 && C % 128 == 0 && att_channels == 128 && Kp == 128;
}

hipError_t launch_asp_fused(const AspFusedParams& p, int B, hipStream_t stream) {
    if (!asp_fused_supported(p.T, p.C, 128, p.Kp) || B <= 0 || p.ldx % 8 != 0) return hipErrorInvalidValue;
    static DeviceOnce attr;
    if (hipError_t e = set_max_dynamic_lds(attr, reinterpret_cast<const void*>(asp_fused_kernel), AF_LDS)) return e;
    hipLaunchKernelGGL(asp_fused_kernel, dim3(B), dim3(256), AF_LDS, stream, p);
    return hipGetLastError();
}

}  // namespace svhip
