// asp_x3.hip — attentive statistics pooling for SVHIP_F32X3 handles: attention logits, softmax over time, weighted mean / std
// and BatchNorm in ONE pass over the mfa output; the (M x 3C) fp32 logits never exist (gfx950).
//
// Reference: AttentiveStatisticsPooling.forward, models/ECAPA_TDNN.py:250-259 and asp_bn (:496):
//   attn = conv1x1(att) + b;  w = softmax_t(attn);  mu = sum_t w x;  sd = sqrt(clamp(sum_t w x^2 - mu^2, 1e-12));  BN(cat[mu, sd])
// On these handles the path used to be a GEMM that wrote 1.26 GB of logits (B = 256) and a pooling kernel that read them back
// next to x: 0.54 + 0.61 ms; this kernel takes 0.46 (round 3: matrix pipe busy 0.28, 275 vector instructions per tile per wave,
// 1.49 GB fetched against 1.26 GB of x + 52 MB of att; x two tiles ahead instead of one changed nothing, a 128-VGPR / two-workgroups-
// per-CU form with ordinary loads spilled and took 0.56).  A workgroup owns (utterance, 256 channels) and walks the utterance in tiles of 32 frames:
//   * a wave owns 32 channels: their asp.conv rows (128 k, bf16 hi | lo planes of the packed S32 weights) are the MFMA B operand,
//     loaded once into 64 VGPRs;
//   * the att tile (32 frames x 128 k fp32, asp.tdnn's output) is split into bf16 hi / lo on its way into LDS (double-buffered,
//     XOR-swizzled 16-byte chunks, one item per thread) and is the A operand of v_mfma_f32_32x32x16_bf16; a logit is
//     hi.hi + hi.lo + lo.hi in fp32 (~2^-17 relative, the x3 error model); the conv bias is dropped (softmax over time is
//     invariant to a per-channel shift);
//   * the C/D layout leaves a lane 16 frames of ONE channel, so the softmax is lane-local: a running maximum per lane, rescaled
//     once per tile, and sum e, sum e d, sum e d^2 with d = x - (the utterance's plain mean of the channel, which the mfa GEMM's
//     column sums already gave): the shift makes var = E_w[d^2] - E_w[d]^2 free of cancellation.  x is read straight from HBM
//     by LDS-DMA into a private ring per wave (128 contiguous bytes per frame), two tiles ahead;
//   * the two lanes of a channel (frame halves) merge their states at the end.
#include "common.h"
#include "kernels.h"

namespace svhip {

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

constexpr int AX_THREADS = 512;
constexpr int AX_PLANE = 32 * 256;                 // one bf16 plane of an att tile: 32 frames x 128 k
constexpr int AX_RAW = 32 * 512;                   // one raw att tile: 32 frames x 128 k fp32
constexpr int AX_XSLAB = 32 * 128;                 // one wave's x tile: 32 frames x 32 channels fp32
constexpr int AX_NSLOT = 3;
constexpr int AX_XRING = 2 * AX_PLANE + 2 * AX_RAW;
constexpr int AX_LDS = AX_XRING + 8 * AX_NSLOT * AX_XSLAB;   // 16 + 32 + 96 KiB
constexpr float AX_L2E = 1.4426950408889634f;

__global__ __launch_bounds__(AX_THREADS, 2) void asp_x3_kernel(AspX3Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* planes = smem;                                                       // hi plane | lo plane of the current att tile
    char* raw = smem + 2 * AX_PLANE;                                           // [2] raw fp32 att tiles (DMA targets)
    // workgroup id -> (utterance, channel group): ids are dealt round-robin to the 8 XCDs, and the C / 256 workgroups of one
    // utterance should share an L2 (they all read its att tiles): XCD x takes utterances x, x + 8, ...
    const int ncg = p.C >> 8;
    int b, cg;
    {
        const int id = blockIdx.x, full = (p.B >> 3) * 8 * ncg;                // ids of whole groups of 8 utterances
        if (id < full) {
            const int slot = id >> 3;
            b = (slot / ncg) * 8 + (id & 7);
            cg = slot - (slot / ncg) * ncg;
        } else {
            const int q = id - full;
            b = (p.B >> 3) * 8 + q / ncg;
            cg = q - (q / ncg) * ncg;
        }
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* xring = smem + AX_XRING + wave * AX_NSLOT * AX_XSLAB;                // this wave's private ring of x tiles
    const int r = lane & 31, h = lane >> 5;
    const int cw = cg * 256 + wave * 32;                               // the wave's first channel
    const int c = cw + r;                                                      // this lane's channel
    const int T = p.T;
    const int nt = (T + 31) >> 5;

    // B operand: channel c, k = 16 ks + 8 h .. + 7 (S32 row: per 32 k, 32 hi | 32 lo)
    bf16x8 wh[8], wl[8];
    {
        const char* wrow = reinterpret_cast<const char*>(p.Ws32) + (int64_t)c * 512;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const char* q = wrow + (ks >> 1) * 128 + ((ks & 1) * 16 + 8 * h) * 2;
            wh[ks] = *reinterpret_cast<const bf16x8*>(q);
            wl[ks] = *reinterpret_cast<const bf16x8*>(q + 64);
        }
    }
    const float mref = p.mref[(int64_t)b * p.mref_ld + c];
    // Every load inside the loop is an LDS-DMA (an ordinary global load beside them makes the compiler drain the queue with
    // vmcnt(0)); rows past the utterance clamp to its last frame and are masked out of the softmax.
    //   att tile: wave w brings frames 4 w .. 4 w + 3 (two instructions of two 512-byte rows)
    //   x tile  : instruction i brings frames 8 i .. 8 i + 7 of the wave's 32 channels (lane: frame 8 i + (lane >> 3), piece lane & 7)
    const char* abase = reinterpret_cast<const char*>(p.att + (int64_t)b * T * 128) + (lane & 31) * 16;
    const char* xbase = reinterpret_cast<const char*>(p.X + (int64_t)b * T * p.ldx + cw) + (lane & 7) * 16;
    const int64_t xrow = (int64_t)p.ldx * 4;
    auto dma_att = [&](int mt, int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = min(mt * 32 + 4 * wave + 2 * i + (lane >> 5), T - 1);
            __builtin_amdgcn_global_load_lds((gbl_void*)(abase + (int64_t)f * 512), (lds_void*)(raw + buf * AX_RAW + (4 * wave + 2 * i) * 512), 16, 0, CPOL_NT);
        }
    };
    auto dma_x = [&](int mt, int slot) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = min(mt * 32 + 8 * i + (lane >> 3), T - 1);
            __builtin_amdgcn_global_load_lds((gbl_void*)(xbase + f * xrow), (lds_void*)(xring + slot * AX_XSLAB + i * 1024), 16, 0, CPOL_NT);
        }
    };
    // raw fp32 tile -> bf16 hi / lo planes, one (frame, 8 k) item per thread, 16-byte chunks XOR-swizzled by the frame
    const int sfr = tid >> 4;
    const int soff = sfr * 256 + (((tid & 15) ^ (sfr & 15)) << 4);
    auto convert = [&](int buf) {
        const f32x4* q = reinterpret_cast<const f32x4*>(raw + buf * AX_RAW + sfr * 512 + (tid & 15) * 32);
        const f32x4 v0 = q[0], v1 = q[1];
        x3x8_t hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = j < 4 ? v0[j] : v1[j - 4];
            const x3_t hb = x3_hi(v);
            hi[j] = hb;
            lo[j] = x3_lo(v, hb);
        }
        *reinterpret_cast<x3x8_t*>(planes + soff) = hi;
        *reinterpret_cast<x3x8_t*>(planes + AX_PLANE + soff) = lo;
    };

    // (__syncthreads() carries a fence that drains vmcnt: LDS traffic only needs lgkmcnt(0) + the raw barrier)
    auto lds_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    float m = -INFINITY, se = 0.0f, s1 = 0.0f, s2 = 0.0f;
    // DMA queue of a wave, oldest first, at the top of iteration mt after its issues:
    //   x(mt) 4 | att(mt+1) 2 | x(mt+1) 4 | att(mt+2) 2 | x(mt+2) 4
    // so x(mt) has landed at vmcnt(12) and att(mt+1) at vmcnt(10) (the counter retires in order).
    dma_att(0, 0);
    dma_x(0, 0);
    dma_att(1, 1);
    dma_x(1, 1);
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");                          // att(0)
    lds_barrier();
    convert(0);
    lds_barrier();
    int slot = 0;
    for (int mt = 0; mt < nt; ++mt) {
        const int fill = slot == 0 ? 2 : slot - 1;                             // (slot + 2) % 3
        dma_att(mt + 2, mt & 1);                                               // raw[mt & 1] was converted an iteration ago
        dma_x(mt + 2, fill);
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int o = r * 256 + (((2 * ks + h) ^ (r & 15)) << 4);
            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(planes + o);
            const bf16x8 al = *reinterpret_cast<const bf16x8*>(planes + AX_PLANE + o);
            acc = X3H::mfma32(al, wh[ks], acc);     // small terms first
            acc = X3H::mfma32(ah, wl[ks], acc);
            acc = X3H::mfma32(ah, wh[ks], acc);
        }
        // lane-local online softmax over this lane's 16 frames of channel c
        if (mt * 32 + 32 > T) {                                                // frames past the utterance: weight 0
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int f = mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                acc[e] = f < T ? acc[e] : -INFINITY;
            }
        }
        float tmax = fmaxf(acc[0], acc[1]);
#pragma unroll
        for (int e = 2; e < 16; e += 2) tmax = fmaxf(fmaxf(acc[e], acc[e + 1]), tmax);  // v_max3
        const float mnew = fmaxf(m, tmax);
        const float mm = mnew == -INFINITY ? 0.0f : mnew;
        const float fs = __builtin_amdgcn_exp2f((m - mm) * AX_L2E);            // m = -inf: 0 (and the sums are 0 anyway)
        se *= fs; s1 *= fs; s2 *= fs;
        const float mb = mm * AX_L2E;
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");                      // x(mt)
        const float* xs = reinterpret_cast<const float*>(xring + slot * AX_XSLAB) + r;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float w = __builtin_amdgcn_exp2f(fmaf(acc[e], AX_L2E, -mb));
            const float d = xs[((e & 3) + 8 * (e >> 2) + 4 * h) * 32] - mref;
            const float wd = w * d;
            se += w;
            s1 += wd;
            s2 = fmaf(wd, d, s2);
        }
        m = mnew;
        slot = slot == 2 ? 0 : slot + 1;
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");                      // att(mt + 1), this wave's rows
        lds_barrier();                                                       // ... everyone's; and the planes are free
        convert((mt + 1) & 1);
        lds_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                           // (clamped prefetches past the last tile)
    // merge the two frame halves of a channel (lanes r and r + 32)
    const float mo = __shfl_xor(m, 32), seo = __shfl_xor(se, 32), s1o = __shfl_xor(s1, 32), s2o = __shfl_xor(s2, 32);
    const float M = fmaxf(m, mo);
    const float MM = M == -INFINITY ? 0.0f : M;
    const float f1 = __builtin_amdgcn_exp2f((m - MM) * AX_L2E), f2 = __builtin_amdgcn_exp2f((mo - MM) * AX_L2E);
    const float SE = se * f1 + seo * f2, S1 = s1 * f1 + s1o * f2, S2 = s2 * f1 + s2o * f2;
    if (h == 0) {
        const float md = S1 / SE;
        const float mean = mref + md;
        const float sd = sqrtf(fmaxf(S2 / SE - md * md, p.eps));
        const int C = p.C;
        if (p.pooled_raw) {
            p.pooled_raw[(int64_t)b * 2 * C + c] = mean;
            p.pooled_raw[(int64_t)b * 2 * C + C + c] = sd;
        }
        p.pooled_bn[(int64_t)b * 2 * C + c] = fmaf(mean, p.bn_scale[c], p.bn_shift[c]);
        p.pooled_bn[(int64_t)b * 2 * C + C + c] = fmaf(sd, p.bn_scale[C + c], p.bn_shift[C + c]);
    }
}

// ---- the same single-pass pooling for bf16 handles: 16 waves per CU instead of asp_fused's 4 (0.195 against 0.264 ms) ----------
// att and x are bf16: the att tile goes to LDS by DMA as it is (swizzled on the source chunk), one MFMA per k step, x tiles of
// 32 frames x 32 channels x 2 bytes per wave through a private ring of three slots; 64 KiB of LDS and < 128 VGPRs: two workgroups
// per CU.  DMA queue of a wave at the end of iteration mt: x(mt + 1) 2 | att(mt + 1) 1 | x(mt + 2) 2 -> vmcnt(2) says att(mt + 1)
// (and the older x(mt + 1)) has landed.
constexpr int AB_ATT = 32 * 256;                   // one att tile: 32 frames x 128 k bf16
constexpr int AB_XSLAB = 32 * 64;                  // one wave's x tile
constexpr int AB_LDS = 2 * AB_ATT + 8 * AX_NSLOT * AB_XSLAB;     // 16 + 48 KiB

__global__ __launch_bounds__(AX_THREADS, 4) void asp_bf16_kernel(AspFusedParams p, const float* __restrict__ mref_all, int mref_ld, int Bn) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ncg = p.C >> 8;
    int b, cg;
    {
        const int id = blockIdx.x, full = (Bn >> 3) * 8 * ncg;
        if (id < full) {
            const int slot = id >> 3;
            b = (slot / ncg) * 8 + (id & 7);
            cg = slot - (slot / ncg) * ncg;
        } else {
            const int q = id - full;
            b = (Bn >> 3) * 8 + q / ncg;
            cg = q - (q / ncg) * ncg;
        }
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* xring = smem + 2 * AB_ATT + wave * AX_NSLOT * AB_XSLAB;
    const int r = lane & 31, h = lane >> 5;
    const int cw = cg * 256 + wave * 32;
    const int c = cw + r;
    const int T = p.T;
    const int nt = (T + 31) >> 5;

    bf16x8 wf[8];
    {
        const char* wrow = reinterpret_cast<const char*>(p.W) + ((int64_t)c * p.Kp + 8 * h) * 2;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) wf[ks] = *reinterpret_cast<const bf16x8*>(wrow + ks * 32);
    }
    const float mref = mref_all[(int64_t)b * mref_ld + c];
    // att: one 16-byte chunk per thread and tile: frame tid >> 4, LDS slot tid & 15 <- source chunk slot ^ (frame & 15)
    const int sfr = tid >> 4;
    const char* abase = reinterpret_cast<const char*>(p.att) + (int64_t)b * T * 256 + (((tid & 15) ^ (sfr & 15)) << 4);
    const char* xbase = reinterpret_cast<const char*>(p.X) + ((int64_t)b * T * p.ldx + cw) * 2 + (lane & 3) * 16;
    const int64_t xrow = (int64_t)p.ldx * 2;
    auto dma_att = [&](int mt, int buf) {
        const int f = min(mt * 32 + sfr, T - 1);
        __builtin_amdgcn_global_load_lds((gbl_void*)(abase + (int64_t)f * 256), (lds_void*)(smem + buf * AB_ATT + wave * 1024), 16, 0, 0);
    };
    auto dma_x = [&](int mt, int slot) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = min(mt * 32 + 16 * i + (lane >> 2), T - 1);
            __builtin_amdgcn_global_load_lds((gbl_void*)(xbase + f * xrow), (lds_void*)(xring + slot * AB_XSLAB + i * 1024), 16, 0, 0);      // (measured: 197 us; with the nt bit 208 — x was just read by gemm_n128 and is found on die)
        }
    };
    auto lds_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    float m = -INFINITY, se = 0.0f, s1 = 0.0f, s2 = 0.0f;
    dma_x(0, 0);
    dma_att(0, 0);
    dma_x(1, 1);
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    lds_barrier();
    int slot = 0;
    for (int mt = 0; mt < nt; ++mt) {
        const int buf = mt & 1;
        const int fill = slot == 0 ? 2 : slot - 1;
        dma_att(mt + 1, buf ^ 1);
        dma_x(mt + 2, fill);
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
        const char* at = smem + buf * AB_ATT + r * 256;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(at + (((2 * ks + h) ^ (r & 15)) << 4));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, wf[ks], acc, 0, 0, 0);
        }
        if (mt * 32 + 32 > T) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int f = mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                acc[e] = f < T ? acc[e] : -INFINITY;
            }
        }
        float tmax = fmaxf(acc[0], acc[1]);
#pragma unroll
        for (int e = 2; e < 16; e += 2) tmax = fmaxf(fmaxf(acc[e], acc[e + 1]), tmax);
        const float mnew = fmaxf(m, tmax);
        const float mm = mnew == -INFINITY ? 0.0f : mnew;
        const float fs = __builtin_amdgcn_exp2f((m - mm) * AX_L2E);
        se *= fs; s1 *= fs; s2 *= fs;
        const float mb = mm * AX_L2E;
        const uint16_t* xs = reinterpret_cast<const uint16_t*>(xring + slot * AB_XSLAB) + r;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float w = __builtin_amdgcn_exp2f(fmaf(acc[e], AX_L2E, -mb));
            const float xv = __uint_as_float((uint32_t)xs[((e & 3) + 8 * (e >> 2) + 4 * h) * 32] << 16);
            const float d = xv - mref;
            const float wd = w * d;
            se += w;
            s1 += wd;
            s2 = fmaf(wd, d, s2);
        }
        m = mnew;
        slot = slot == 2 ? 0 : slot + 1;
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");       // att(mt + 1) (this wave's rows) and x(mt + 1)
        lds_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const float mo = __shfl_xor(m, 32), seo = __shfl_xor(se, 32), s1o = __shfl_xor(s1, 32), s2o = __shfl_xor(s2, 32);
    const float M = fmaxf(m, mo);
    const float MM = M == -INFINITY ? 0.0f : M;
    const float f1 = __builtin_amdgcn_exp2f((m - MM) * AX_L2E), f2 = __builtin_amdgcn_exp2f((mo - MM) * AX_L2E);
    const float SE = se * f1 + seo * f2, S1 = s1 * f1 + s1o * f2, S2 = s2 * f1 + s2o * f2;
    if (h == 0) {
        const float md = S1 / SE;
        const float mean = mref + md;
        const float sd = sqrtf(fmaxf(S2 / SE - md * md, p.eps));
        const int C = p.C;
        if (p.pooled_raw) {
            p.pooled_raw[(int64_t)b * 2 * C + c] = mean;
            p.pooled_raw[(int64_t)b * 2 * C + C + c] = sd;
        }
        p.pooled_bn[(int64_t)b * 2 * C + c] = fmaf(mean, p.bn_scale[c], p.bn_shift[c]);
        p.pooled_bn[(int64_t)b * 2 * C + C + c] = fmaf(sd, p.bn_scale[C + c], p.bn_shift[C + c]);
    }
}

}  // namespace

hipError_t launch_asp_bf16(const AspFusedParams& p, const float* mref, int mref_ld, int B, hipStream_t stream) {
    if (p.T < 1 || p.C % 256 != 0 || p.Kp != 128 || !p.att || !p.W || !p.X || !mref || !p.pooled_bn || B <= 0 || p.ldx % 8 != 0) return hipErrorInvalidValue;
    static DeviceOnce once;
    hipError_t e = set_max_dynamic_lds(once, reinterpret_cast<const void*>(asp_bf16_kernel), AB_LDS);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(asp_bf16_kernel, dim3((p.C / 256) * B), dim3(AX_THREADS), AB_LDS, stream, p, mref, mref_ld, B);
    return hipGetLastError();
}

bool asp_x3_supported(int T, int C, int att_channels, int K) {
    return T >= 1 && C % 256 == 0 && att_channels == 128 && K == 128;
}

hipError_t launch_asp_x3(const AspX3Params& p, int B, hipStream_t stream) {
    if (!asp_x3_supported(p.T, p.C, 128, 128) || !p.att || !p.Ws32 || !p.X || !p.mref || !p.pooled_bn || B <= 0) return hipErrorInvalidValue;
    if (((reinterpret_cast<uintptr_t>(p.att) | reinterpret_cast<uintptr_t>(p.Ws32)) & 15) != 0) return hipErrorInvalidValue;
    static DeviceOnce once;
    hipError_t e = set_max_dynamic_lds(once, reinterpret_cast<const void*>(asp_x3_kernel), AX_LDS);
    if (e != hipSuccess) return e;
    AspX3Params q = p;
    q.B = B;
    hipLaunchKernelGGL(asp_x3_kernel, dim3((p.C / 256) * B), dim3(AX_THREADS), AX_LDS, stream, q);
    return hipGetLastError();
}

}  // namespace svhip
