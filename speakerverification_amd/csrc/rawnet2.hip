// rawnet2.hip — RawNet2 (sinc front-end, ASP aggregation) kernels that are not plain GEMMs.
// Reference: models/RawNet2_custom.py:161-227, models/RawNet_baseline.py:13-24 (LayerNorm),
// :62-68 (AFMS), :221-232 (RawNetBasicBlock), :320-361 (SincConv_fast).
//
//   rn_ln_stats   : per-utterance mean and 1/(unbiased std + 1e-6) over the L samples
//   rn_sinc       : LayerNorm apply + sinc conv (k = 251, valid) + |.| + max_pool1d(3) + BN + LeakyReLU(0.3)
//                   in ONE kernel: the (B, 128, 31750) conv output never exists.  The conv is an MFMA
//                   product whose "im2col" operand is read straight out of an LDS copy of the waveform
//                   (hop 1 => rows overlap in all but one sample).  MFMA tile (g, j) holds conv positions
//                   3*(32g + r) + j, so the three max-pool partners of a pooled frame sit in the same
//                   lane / register of three accumulators and the pool is an elementwise max.
//                   16-bit: v_mfma_f32_16x16x32 (round 4; 32x32x16 before: 477 -> 460 us), operand fetched as aligned 16-byte
//                   reads from 8 sample-shifted LDS copies;  fp32: v_mfma_f32_32x32x2_f32 with 4-byte reads.
//   rn_bn_act     : y = LeakyReLU(x*scale + shift)                       (pre-activation of the blocks)
//   rn_maxpool3   : max_pool1d(3) along frames
//   rn_afms_apply : (x + alpha) * s[b, c]                                (AFMS, RawNet_baseline.py:66-67)
//   rn_attn_pool  : softmax over frames, m = sum x w, s = sqrt(clamp(sum x^2 w - m^2, 1e-5))
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace svhip {

namespace {

// With `xn` the kernel also writes the LayerNorm output gamma * (x - mean) * inv + beta (RawNet_baseline.py:24) as bf16 into a
// zero-tailed row of Lp samples: the bf16 sinc kernel stages its operand from it with LDS-DMA (no arithmetic in its tile loop).
// LO (the split front-end of F32X3 handles): four rows per utterance — hi parts of the two copies, then their lo parts (lo = H(v - hi))
template <typename H, bool LO = false>
__global__ __launch_bounds__(256) void rn_ln_stats_kernel(const float* __restrict__ x, int L, float* __restrict__ stats,
                                                          H* __restrict__ xn, int Lp, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    const float* __restrict__ p = x + (int64_t)b * L;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // 16-byte loads, eight in flight per thread (scalar loads, one per trip, ran this kernel at 0.4 TB/s); L % 4 tail scalar
    const int L4 = ((reinterpret_cast<uintptr_t>(p) & 15) == 0) ? (L >> 2) : 0;
    const f32x4* __restrict__ p4 = reinterpret_cast<const f32x4*>(p);
    float s = 0.0f;
#pragma unroll 8
    for (int i = threadIdx.x; i < L4; i += 256) { const f32x4 v = p4[i]; s += (v[0] + v[1]) + (v[2] + v[3]); }
    for (int i = 4 * L4 + threadIdx.x; i < L; i += 256) s += p[i];
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)L;
    __syncthreads();
    float q = 0.0f;
#pragma unroll 8
    for (int i = threadIdx.x; i < L4; i += 256) {
        const f32x4 v = p4[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[e] - mean; q = fmaf(d, d, q); }
    }
    for (int i = 4 * L4 + threadIdx.x; i < L; i += 256) { const float d = p[i] - mean; q = fmaf(d, d, q); }
    q = wave_sum(q);
    if (lane == 0) red[wave] = q;
    __syncthreads();
    const float var = (red[0] + red[1] + red[2] + red[3]) / (float)(L - 1);          // torch.std: unbiased
    const float inv = 1.0f / (sqrtf(var) + 1e-6f);
    if (threadIdx.x == 0) {
        stats[2 * b] = mean;
        stats[2 * b + 1] = inv;
    }
    if (xn) {
        // copy 0: sample j at index j; copy 1: sample j + 1 at index j — so that every 2-sample LDS-DMA of the sinc kernel, for an
        // even or an odd shift, starts on a 4-byte boundary
        H* __restrict__ o0 = xn + (int64_t)b * (LO ? 4 : 2) * Lp;
        H* __restrict__ o1 = o0 + Lp;
        const bool vec = (L % 8 == 0) && ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta)) & 15) == 0;
        for (int j0 = threadIdx.x * 8; j0 < Lp; j0 += 256 * 8) {
            float v[9];
            if (vec && j0 + 8 <= L) {                                // 16-byte loads (scalar loads ran this pass at 50 us per launch)
                const f32x4 x0 = *reinterpret_cast<const f32x4*>(p + j0), x1 = *reinterpret_cast<const f32x4*>(p + j0 + 4);
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + j0), g1 = *reinterpret_cast<const f32x4*>(gamma + j0 + 4);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + j0), b1 = *reinterpret_cast<const f32x4*>(beta + j0 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = g0[e] * (x0[e] - mean) * inv + b0[e];
                    v[4 + e] = g1[e] * (x1[e] - mean) * inv + b1[e];
                }
                v[8] = j0 + 8 < L ? gamma[j0 + 8] * (p[j0 + 8] - mean) * inv + beta[j0 + 8] : 0.0f;
            } else {
#pragma unroll
                for (int e = 0; e < 9; ++e) {
                    const int j = j0 + e;
                    v[e] = j < L ? gamma[j] * (p[j] - mean) * inv + beta[j] : 0.0f;
                }
            }
            Vec16<H> a, c;
#pragma unroll
            for (int e = 0; e < 8; ++e) { a.set(e, v[e]); c.set(e, v[e + 1]); }
            *reinterpret_cast<Vec16<H>*>(o0 + j0) = a;
            *reinterpret_cast<Vec16<H>*>(o1 + j0) = c;
            if (LO) {
                Vec16<H> al, cl;
#pragma unroll
                for (int e = 0; e < 8; ++e) { al.set(e, v[e] - a.get(e)); cl.set(e, v[e + 1] - c.get(e)); }
                *reinterpret_cast<Vec16<H>*>(o0 + 2 * Lp + j0) = al;
                *reinterpret_cast<Vec16<H>*>(o1 + 2 * Lp + j0) = cl;
            }
        }
    }
}

// The same, with the utterance held in REGISTERS (round 5): 1 024 threads, thread t owns the 8-sample units t, t + 1024, ... (NU of them: L <= 8 192 NU),
// so the waveform is read from memory ONCE (the kernel above walks it three times — mean, variance, output — with four waves on one CU:
// three dependent 128 KB passes, 34 us at B = 256, a launch that is pure latency) and sixteen waves keep its loads in flight.
// Same formulas; the sums are formed per thread, per wave, then over the sixteen waves in order (another order than above: fp32 round-off).
template <typename H, bool LO, int NU>
__global__ __launch_bounds__(1024) void rn_ln_stats_reg_kernel(const float* __restrict__ x, int L, float* __restrict__ stats,
                                                               H* __restrict__ xn, int Lp, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta) {
    __shared__ float red[16];
    const int b = blockIdx.x;
    const float* __restrict__ p = x + (int64_t)b * L;
    const f32x4* __restrict__ p4 = reinterpret_cast<const f32x4*>(p);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nunits = L >> 3;                                  // (L % 8 == 0, rows 16-byte aligned: host check)
    f32x4 r[NU][2];
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < NU; ++k) {
        const int u = threadIdx.x + 1024 * k;
        if (u < nunits) {
            r[k][0] = p4[2 * u];
            r[k][1] = p4[2 * u + 1];
        } else {
            r[k][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            r[k][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
#pragma unroll
    for (int k = 0; k < NU; ++k) s += ((r[k][0][0] + r[k][0][1]) + (r[k][0][2] + r[k][0][3])) + ((r[k][1][0] + r[k][1][1]) + (r[k][1][2] + r[k][1][3]));
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    float tot = 0.0f;
#pragma unroll
    for (int w = 0; w < 16; ++w) tot += red[w];
    const float mean = tot / (float)L;
    __syncthreads();
    float q = 0.0f;
#pragma unroll
    for (int k = 0; k < NU; ++k) {
        if (threadIdx.x + 1024 * k < nunits) {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = r[k][h2][e] - mean; q = fmaf(d, d, q); }
        }
    }
    q = wave_sum(q);
    if (lane == 0) red[wave] = q;
    __syncthreads();
    tot = 0.0f;
#pragma unroll
    for (int w = 0; w < 16; ++w) tot += red[w];
    const float var = tot / (float)(L - 1);                      // torch.std: unbiased
    const float inv = 1.0f / (sqrtf(var) + 1e-6f);
    if (threadIdx.x == 0) {
        stats[2 * b] = mean;
        stats[2 * b + 1] = inv;
    }
    if (xn) {
        H* __restrict__ o0 = xn + (int64_t)b * (LO ? 4 : 2) * Lp;
        H* __restrict__ o1 = o0 + Lp;
#pragma unroll
        for (int k = 0; k < NU; ++k) {
            const int u = threadIdx.x + 1024 * k;
            const int j0 = 8 * u;
            if (j0 >= Lp) continue;
            float v[9];
            if (u < nunits) {
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + j0), g1 = *reinterpret_cast<const f32x4*>(gamma + j0 + 4);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + j0), b1 = *reinterpret_cast<const f32x4*>(beta + j0 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = g0[e] * (r[k][0][e] - mean) * inv + b0[e];
                    v[4 + e] = g1[e] * (r[k][1][e] - mean) * inv + b1[e];
                }
                v[8] = j0 + 8 < L ? gamma[j0 + 8] * (p[j0 + 8] - mean) * inv + beta[j0 + 8] : 0.0f;
            } else {
#pragma unroll
                for (int e = 0; e < 9; ++e) v[e] = 0.0f;
            }
            Vec16<H> a, c;
#pragma unroll
            for (int e = 0; e < 8; ++e) { a.set(e, v[e]); c.set(e, v[e + 1]); }
            *reinterpret_cast<Vec16<H>*>(o0 + j0) = a;
            *reinterpret_cast<Vec16<H>*>(o1 + j0) = c;
            if (LO) {
                Vec16<H> al, cl;
#pragma unroll
                for (int e = 0; e < 8; ++e) { al.set(e, v[e] - a.get(e)); cl.set(e, v[e + 1] - c.get(e)); }
                *reinterpret_cast<Vec16<H>*>(o0 + 2 * Lp + j0) = al;
                *reinterpret_cast<Vec16<H>*>(o1 + 2 * Lp + j0) = cl;
            }
        }
    }
}

constexpr int SINC_PT = 64;                 // pooled frames per iteration
constexpr int SINC_SAMPLES = 464;           // >= 3 * SINC_PT conv positions + 255 + 8, multiple of 8

template <typename T> struct SincCfg;
struct SincCfg16 {
    // Copy c starts at c * COPY_BYTES + 16 * ((COPY_SKEW >> 4c) & 15): start offsets (mod 256 bytes) found by search so that the
    // ds_read_b128 lane groups of the fragment pattern (positions 3 * frame + j: stride-3 rows across the 8 copies) spread over
    // the banks.  A uniform stride of 960 bytes cost one extra LDS cycle per lane group on average (PMC: bank-conflict cycles
    // = 28 % of the kernel's CU cycles); this assignment a third of that — no assignment of 8 offsets is conflict-free.
    // (round 4: the fragment pattern is that of v_mfma_f32_16x16x32 — 16 lanes = 16 pooled frames at one k chunk, conv positions 3 frame + j —
    //  and the offsets were searched again for it: exhaustively, no assignment is conflict-free; this one has one 2-way pair in two of the
    //  three pool-partner patterns)
    static constexpr int COPY_BYTES = 1280;                    // 928 bytes of samples + up to 240 of skew, 256-byte multiple
    static constexpr unsigned COPY_SKEW = 0x7E49C210u;         // offsets 0, 1, 2, 12, 9, 4, 14, 7 (x 16 bytes) for copies 0..7
    static constexpr int XLDS = 8 * COPY_BYTES;                // one operand buffer: 8 sample-shifted copies of the tile
    static constexpr int OUT_OFF = (2 * XLDS + 255) & ~255;    // two operand buffers, then two output images
    static constexpr int OUT_BYTES = SINC_PT * 256;            // output tile: SINC_PT pooled frames x 128 filters, bf16
    static constexpr int BN_OFF = OUT_OFF + 2 * OUT_BYTES;     // first_bn scale[128], shift[128] (fp32)
    static constexpr int LDS = BN_OFF + 1024;
};
template <> struct SincCfg<bf16_t> : SincCfg16 {};
template <> struct SincCfg<f16_t> : SincCfg16 {};
template <> struct SincCfg<float> {
    static constexpr int COPY_BYTES = 0;
    static constexpr unsigned COPY_SKEW = 0;
    static constexpr int XLDS = SINC_SAMPLES * 4;
    static constexpr int OUT_OFF = 0;
    static constexpr int OUT_BYTES = 0;
    static constexpr int BN_OFF = SINC_SAMPLES * 4;
    static constexpr int LDS = BN_OFF + 1024;
};

// filters: bf16: [128][256] bf16 (k contiguous); fp32: [128][252] fp32.
// SYM (round 6, fp16 handles): the sinc band-pass filters are symmetric about their centre tap (RawNet_baseline.py:339-357 builds the right
// half as the flip of the left), so  y[s] = sum_{m = 0..125} h[125 + m] (x[c + m] + x[c - m]),  c = s + 125  (the centre weight halved):
// K = 126 instead of 251 — HALF the MFMAs and half the filter registers.  The operand fragment of (position s, slots k' .. k' + 7) is formed
// in registers from TWO fragment reads of the same sample-shifted copies: the forward window x[c - 2 + k' ..] and the backward window
// x[c + 2 - k' - 7 ..], added with the second one's eight halves reversed (v_pk_add_f16 with swapped half selects: no extra instruction).
// Slot k' stands for m = k' - 2; slots 0 and 1 carry zero weights, so that every read stays inside the tile's 464 samples.
// filters: [128][128] fp16, slot-major.
template <typename T, bool SYM = false>
__global__ __launch_bounds__(256, 2) void rn_sinc_kernel(const float* __restrict__ wav, const float* __restrict__ stats,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const void* __restrict__ filt, const float* __restrict__ bn_scale,
                                                         const float* __restrict__ bn_shift, T* __restrict__ out, T* __restrict__ pre, const float* __restrict__ nscale,
                                                         const float* __restrict__ nshift, int L, int T1,
                                                         int B, const uint16_t* __restrict__ xn, int Lp) {
    typedef SincCfg<T> CF;
    constexpr bool BF = sizeof(T) == 2;
    typedef typename std::conditional<BF, T, bf16_t>::type H;      // the 16-bit operand type (bf16 or fp16); unused on the fp32 path
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // Persistent: the grid is two workgroups per CU, each owns a contiguous range of (utterance, tile) items and loads its filter
    // fragments ONCE (32 KiB per wave, every wave of every workgroup from the same 64 KiB table: with 4 tiles per workgroup and
    // 10 752 workgroups that start-up was ~5 us each, a fifth of the kernel — 723 / 613 / 553 / 497 us at 2 / 4 / 8 / 42 tiles).
    const int tiles_u = (T1 + SINC_PT - 1) / SINC_PT;
    const int items = B * tiles_u;                               // (< 2^31: checked by the launcher)
    const int per = (items + (int)gridDim.x - 1) / (int)gridDim.x;
    const int item0 = (int)blockIdx.x * per, item1 = item0 + per < items ? item0 + per : items;
    if (item0 >= items) return;                                  // workgroup-uniform
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // fp32: wave = n-tile (filters wave*32 .. +31), both position groups.  bf16: wave = (filter half nf, position group gw):
    // 64 filters x 32 pooled positions, so every operand fragment read from LDS feeds TWO MFMAs — with one filter block per
    // wave the kernel sat on the LDS read port (1 KiB per MFMA from 8 waves = 256 B/clk per CU): 0.91 ms -> see DESIGN.md
    const int nf = wave & 1, gw = wave >> 1;
    const int fr = lane & 31, fh = lane >> 5;          // fp32 path: 32x32x2 fragment coordinates
    const int r16 = lane & 15, q4 = lane >> 4;         // 16-bit path: 16x16x32 fragment coordinates

    // this wave's filter fragments stay in registers for the whole kernel (weights are the MFMA A operand).  16-bit path (round 4):
    // v_mfma_f32_16x16x32 — four blocks of 16 filters x eight k steps of 32 (the chip holds a higher clock under this instruction than under
    // 32x32x16: tools/mfma_rate.hip); a lane owns filters 4 q4 .. + 3 of a block for pooled frame r16 of a 16-frame group
    static_assert(!SYM || std::is_same<T, f16_t>::value, "the symmetric form adds operand halves with v_pk_add_f16");
    constexpr int NKS = SYM ? 4 : 8;                   // k steps of 32 slots
    // SYM: a wave owns ALL 128 filters (eight blocks of 16) for 16 pooled frames — its operand costs two fragment reads per k step, so it
    // must feed eight MFMAs, not four (with four the kernel sat on the LDS read port: 386 us against 462 for the 251-tap form)
    constexpr int NFB = SYM ? 8 : 4;
    bf16x8 wfb[BF ? NFB : 1][BF ? NKS : 1];
    float wff[BF ? 1 : 126];
    if (BF) {
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb) {
            const char* w = reinterpret_cast<const char*>(filt) + (int64_t)((SYM ? 0 : nf * 64) + fb * 16 + r16) * (NKS * 32) * 2 + q4 * 16;
#pragma unroll
            for (int kk = 0; kk < NKS; ++kk) wfb[fb][kk] = *reinterpret_cast<const bf16x8*>(w + kk * 64);
        }
    } else {
        const float* w = reinterpret_cast<const float*>(filt) + (int64_t)(wave * 32 + fr) * 252 + fh;
#pragma unroll
        for (int kk = 0; kk < 126; ++kk) wff[kk] = w[2 * kk];
    }

    // bf16: no arithmetic and no VGPR traffic on the way in — the LayerNorm output exists as bf16 (rn_ln_stats), and the 8
    // sample-shifted copies of the NEXT tile are written by 2-byte LDS-DMA while this tile's MFMAs run (two operand buffers).
    // The tile's output goes through an LDS image (two of them) and leaves as whole 256-byte rows one iteration later.
    // History: register-staged operand + 16-byte scattered stores 720 us (ablations: no stores 460, no MFMAs 408) -> samples
    // requested one tile ahead + LDS output image 637 us -> LDS-DMA operand, one barrier per tile: see DESIGN.md.
    auto flush = [&](int b, int tpf, const char* img) {              // 64 rows x 16 chunks of 16 bytes, 4 per thread
        int tid_f = tid;
        asm volatile("" : "+v"(tid_f));                              // (per-call address arithmetic: hoisted out of the tile loop it spills)
#pragma unroll
        for (int e = 0; e < SINC_PT * 16 / 256; ++e) {
            const int idx = e * 256 + tid_f;
            const int row = idx >> 4, c16 = idx & 15;
            const u32x4 t = *reinterpret_cast<const u32x4*>(img + row * 256 + ((c16 ^ ((row >> 1) & 15)) << 4));
            const u32x4 d = (row & 1) ? u32x4{t[2], t[3], t[0], t[1]} : t;
            if (tpf + row < T1) *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(out) + (((int64_t)b * T1 + tpf + row) * 128 + c16 * 8) * 2) = d;
        }
    };
    typedef __attribute__((address_space(3))) void lds_void_t;
    typedef __attribute__((address_space(1))) const void gbl_void_t;
    auto dma = [&](int item, int buf) {                          // copy pc, index i <- sample 3 * tpn + i + pc of utterance b
        const int b = item / tiles_u, tpn = (item - b * tiles_u) * SINC_PT;
        // 4-byte DMAs (two samples per lane, LDS destination = wave-uniform base + 4 * lane): the source of an odd shift comes
        // from the second, one-sample-shifted copy of the waveform, so every source address is 4-byte aligned
        const uint16_t* src0 = xn + (int64_t)b * 2 * Lp + 3 * tpn + 2 * lane;
        char* dst = smem + buf * CF::XLDS;
#pragma unroll
        for (int e = 0; e < 8; ++e) {                                // 8 copies x 4 segments of 128 samples = 32 wave-instructions, 8 per wave
            const int idx = wave + 4 * e, pc = idx >> 2, q = idx & 3;
            const uint16_t* src = src0 + (pc & 1) * Lp + (pc & ~1) + q * 128;
            if (q * 128 + 2 * lane < SINC_SAMPLES)
                __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(dst + pc * CF::COPY_BYTES + ((CF::COPY_SKEW >> (4 * pc)) & 15) * 16 + q * 256), 4, 0, 0);
        }
    };
    constexpr int NSMP = (SINC_SAMPLES + 255) / 256;
    float xr[NSMP];                                                 // fp32 path: raw samples of the NEXT tile (gamma / beta are L2-resident)
    auto fetch = [&](int item) {
        if (item >= item1) return;
        const int b = item / tiles_u, tpn = (item - b * tiles_u) * SINC_PT;
        const float* __restrict__ x = wav + (int64_t)b * L;
        int tid_l = tid;
        asm volatile("" : "+v"(tid_l));
#pragma unroll
        for (int e = 0; e < NSMP; ++e) {
            const int j = 3 * tpn + tid_l + 256 * e;
            const bool ok = tid_l + 256 * e < SINC_SAMPLES && j < L;
            xr[e] = x[ok ? j : 0];
        }
    };
    // the BatchNorm constants of the epilogue live in LDS: an ordinary global load inside the tile loop makes the compiler drain
    // vmcnt — the next tile's DMAs and the previous tile's row stores — before the epilogue can start
    float* const bnl = reinterpret_cast<float*>(smem + CF::BN_OFF);
    bnl[tid] = tid < 128 ? bn_scale[tid] : bn_shift[tid - 128];
    if (BF) dma(item0, 0);
    else fetch(item0);
    int tp_prev = -1, b_prev = 0;
    for (int item = item0; item < item1; ++item) {
        const int it = item - item0;
        const int b = item / tiles_u;
        const int tp0 = (item - b * tiles_u) * SINC_PT;
        const int s0 = 3 * tp0;                                      // first sample of this tile
        // bf16: this tile's operand (DMA issued one iteration ago; the barrier's vmcnt(0) retires it) is in buffer it & 1, every
        // wave is past the MFMAs of the previous tile and past its epilogue (output image (it - 1) & 1 complete)
        if (BF) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        char* const xbuf = smem + (BF ? (it & 1) * CF::XLDS : 0);
        char* const otile = smem + CF::OUT_OFF + (it & 1) * CF::OUT_BYTES;
        if (BF) {
            if (item + 1 < item1) dma(item + 1, (it + 1) & 1);
            if (tp_prev >= 0) flush(b_prev, tp_prev, smem + CF::OUT_OFF + ((it - 1) & 1) * CF::OUT_BYTES);
        } else {
            // this tile's samples were requested one iteration ago (before the previous tile's MFMAs): no global round trip here
            const float mean = stats[2 * b], inv = stats[2 * b + 1];
            int tid_s = tid;
            asm volatile("" : "+v"(tid_s));
#pragma unroll
            for (int e = 0; e < NSMP; ++e) {
                const int i = tid_s + 256 * e;
                if (i < SINC_SAMPLES) {
                    const int j = s0 + i;
                    reinterpret_cast<float*>(smem)[i] = j < L ? gamma[j] * (xr[e] - mean) * inv + beta[j] : 0.0f;      // RawNet_baseline.py:24
                }
            }
            fetch(item + 1);                                         // next tile's samples: in flight under this tile's MFMAs
            __syncthreads();
        }
        // acc[a][j]: fp32: a = position group g (filters of this wave); bf16: a = filter block of this wave's half (position group gw)
        f32x16 acc[BF ? 1 : 2][BF ? 1 : 3];           // fp32 path: [position group][pool partner]
        // 16-bit path: the wave's 32 pooled frames are two groups of 16, multiplied and finished ONE AFTER THE OTHER (12 accumulators of
        // 16 x 16 live at a time: with all 24 the kernel needs 270 registers and spills its filter fragments into the tile loop)
        const char* base[3] = {nullptr, nullptr, nullptr};
        const char* bback[3] = {nullptr, nullptr, nullptr};       // SYM: the backward windows (descending with the k step)
        if (BF) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int s = 3 * ((SYM ? 16 * wave : 32 * gw) + r16) + j;       // conv position inside the tile (frame group 0; group 1: + 48)
                if (SYM) {
                    // (slot k': forward sample s + 123 + k', backward sample s + 127 - k': formed per pool partner in the tile body below)
                } else {
                    base[j] = xbuf + (s & 7) * CF::COPY_BYTES + (((CF::COPY_SKEW >> (4 * (s & 7))) & 15) + (s >> 3) + q4) * 16;
                }
            }
        } else {
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[g][j][r] = 0.0f;
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const float* base = reinterpret_cast<const float*>(smem) + 3 * (32 * g + fr) + j + fh;
#pragma unroll
                    for (int kk = 0; kk < 126; ++kk)
                        acc[g][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wff[kk], base[2 * kk], acc[g][j], 0, 0, 0);
                }
        }
        // |.| -> max over the 3 pool partners -> BN -> LeakyReLU(0.3); lane = pooled frame, 4 consecutive filters per accumulator
        int fr_e = BF ? r16 : fr, fh_e = BF ? q4 : fh;
        asm volatile("" : "+v"(fr_e), "+v"(fh_e));
        if (SYM) {
            // pool partner j outermost: eight accumulators (one per filter block) live at a time, folded into a running max of |.|.  The twelve
            // (partner, k step) fragment pairs form ONE read-ahead chain: the next partner's first pair is requested under this partner's last
            // MFMAs (its two window bases are formed there, from the lane coordinates: kept across the tile they cost the registers the filter
            // fragments need)
            f32x4 mx[8];
            auto bases = [&](int j, const char*& fb_, const char*& bb_) {
                int r16_j = r16, q4_j = q4;
                asm volatile("" : "+v"(r16_j), "+v"(q4_j));
                const int sj = 3 * (16 * wave + r16_j) + j, sf = sj + 123, sb = sj + 120;
                fb_ = xbuf + (sf & 7) * CF::COPY_BYTES + (((CF::COPY_SKEW >> (4 * (sf & 7))) & 15) + (sf >> 3) + q4_j) * 16;
                bb_ = xbuf + (sb & 7) * CF::COPY_BYTES + (((CF::COPY_SKEW >> (4 * (sb & 7))) & 15) + (sb >> 3) - q4_j) * 16;
            };
            const char *fbase, *bbase;
            bases(0, fbase, bbase);
            bf16x8 fwd[2], bwd[2];
            fwd[0] = *reinterpret_cast<const bf16x8*>(fbase);
            bwd[0] = *reinterpret_cast<const bf16x8*>(bbase);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                f32x4 a8[8];
#pragma unroll
                for (int fb = 0; fb < 8; ++fb) a8[fb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int cur = (4 * j + kk) & 1;
                    if (kk + 1 < 4) {
                        fwd[cur ^ 1] = *reinterpret_cast<const bf16x8*>(fbase + (kk + 1) * 64);
                        bwd[cur ^ 1] = *reinterpret_cast<const bf16x8*>(bbase - (kk + 1) * 64);
                    } else if (j + 1 < 3) {
                        bases(j + 1, fbase, bbase);
                        fwd[cur ^ 1] = *reinterpret_cast<const bf16x8*>(fbase);
                        bwd[cur ^ 1] = *reinterpret_cast<const bf16x8*>(bbase);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const f16x8 fw = __builtin_bit_cast(f16x8, fwd[cur]), bk = __builtin_bit_cast(f16x8, bwd[cur]);
                    const bf16x8 xf = __builtin_bit_cast(bf16x8, fw + __builtin_shufflevector(bk, bk, 7, 6, 5, 4, 3, 2, 1, 0));      // e[u] = forward[u] + backward[7 - u]
#pragma unroll
                    for (int fb = 0; fb < 8; ++fb) a8[fb] = Half16<H>::mfma16(wfb[fb][kk], xf, a8[fb]);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int fb = 0; fb < 8; ++fb)
#pragma unroll
                    for (int e = 0; e < 4; ++e) mx[fb][e] = j == 0 ? fabsf(a8[fb][e]) : fmaxf(mx[fb][e], fabsf(a8[fb][e]));
            }
            const int row = 16 * wave + fr_e;
            const int tp = tp0 + row;
#pragma unroll
            for (int fb = 0; fb < 8; ++fb) {
                const int f = fb * 16 + 4 * fh_e;
                const f32x4 sc = *reinterpret_cast<const f32x4*>(bnl + f);
                const f32x4 sh = *reinterpret_cast<const f32x4*>(bnl + 128 + f);
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float y = fmaf(mx[fb][e], sc[e], sh[e]); v[e] = y > 0.0f ? y : 0.3f * y; }
                if (tp < T1) {
                    typedef H h16x4 __attribute__((ext_vector_type(4)));
                    const h16x4 pk = {static_cast<H>(v[0]), static_cast<H>(v[1]), static_cast<H>(v[2]), static_cast<H>(v[3])};
                    const int c8 = f >> 2;
                    *reinterpret_cast<h16x4*>(otile + row * 256 + ((c8 ^ (row & 31)) << 3)) = pk;
                    if (pre) {
                        const f32x4 nsc = *reinterpret_cast<const f32x4*>(nscale + f), nsh = *reinterpret_cast<const f32x4*>(nshift + f);
                        h16x4 pp;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { const float t = fmaf(static_cast<float>(pk[e]), nsc[e], nsh[e]); pp[e] = static_cast<H>(t > 0.0f ? t : 0.3f * t); }
                        *reinterpret_cast<h16x4*>(pre + ((int64_t)b * T1 + tp) * 128 + f) = pp;
                    }
                }
            }
        } else
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            // 16-bit path: g = 16-frame group of this wave's 32 pooled frames, q = filter block; fp32 path: g = position group, q = 8-filter group
            f32x4 acc16[BF ? 4 : 1][BF ? 3 : 1];              // [filter block][pool partner]
            if (BF) {
#pragma unroll
                for (int fb = 0; fb < 4; ++fb)
#pragma unroll
                    for (int j = 0; j < 3; ++j) acc16[fb][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                // K step outermost: 12 independent accumulators; an operand fragment (one ds_read_b128) feeds the four filter blocks.  The
                // 24 fragment reads are issued two ahead of their MFMAs through a ring of three, fenced (the compiler otherwise hoists them
                // in bulk); frame group 1 starts 48 conv positions = 6 chunks of the SAME copy further on (+ 96 bytes)
                auto xread = [&](int idx) { return *reinterpret_cast<const bf16x8*>(base[idx % 3] + g * 96 + (idx / 3) * 64); };
                auto bread = [&](int idx) { return *reinterpret_cast<const bf16x8*>(bback[idx % 3] + g * 96 - (idx / 3) * 64); };
                bf16x8 ring[3], rback[SYM ? 3 : 1];
                ring[0] = xread(0);
                ring[1] = xread(1);
                if (SYM) { rback[0] = bread(0); rback[1] = bread(1); }
#pragma unroll
                for (int idx = 0; idx < 3 * NKS; ++idx) {
                    if (idx + 2 < 3 * NKS) {
                        ring[(idx + 2) % 3] = xread(idx + 2);
                        if (SYM) rback[(idx + 2) % 3] = bread(idx + 2);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    bf16x8 xf = ring[idx % 3];
                    if (SYM) {          // e[u] = forward[u] + backward[7 - u]
                        const f16x8 fw = __builtin_bit_cast(f16x8, ring[idx % 3]), bk = __builtin_bit_cast(f16x8, rback[idx % 3]);
                        xf = __builtin_bit_cast(bf16x8, fw + __builtin_shufflevector(bk, bk, 7, 6, 5, 4, 3, 2, 1, 0));
                    }
#pragma unroll
                    for (int fb = 0; fb < 4; ++fb) acc16[fb][idx % 3] = Half16<H>::mfma16(wfb[fb][idx / 3], xf, acc16[fb][idx % 3]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            const int tp = tp0 + (BF ? 32 * gw + 16 * g : 32 * g) + fr_e;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int f = BF ? nf * 64 + q * 16 + 4 * fh_e : wave * 32 + 8 * q + 4 * fh_e;
                const f32x4 sc = *reinterpret_cast<const f32x4*>(bnl + f);
                const f32x4 sh = *reinterpret_cast<const f32x4*>(bnl + 128 + f);
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float m;
                    if (BF) m = fmaxf(fmaxf(fabsf(acc16[q][0][e]), fabsf(acc16[q][1][e])), fabsf(acc16[q][2][e]));
                    else { const int r = 4 * q + e; m = fmaxf(fmaxf(fabsf(acc[g][0][r]), fabsf(acc[g][1][r])), fabsf(acc[g][2][r])); }
                    const float y = fmaf(m, sc[e], sh[e]);
                    v[e] = y > 0.0f ? y : 0.3f * y;
                }
                if (tp < T1) {
                    const int64_t oi = ((int64_t)b * T1 + tp) * 128 + f;
                    // second output: block 0's pre-activation lrelu(bn1(x)), from x as stored (== a separate rn_bn_act pass)
                    f32x4 nsc = {0.f, 0.f, 0.f, 0.f}, nsh = {0.f, 0.f, 0.f, 0.f};
                    if (pre) { nsc = *reinterpret_cast<const f32x4*>(nscale + f); nsh = *reinterpret_cast<const f32x4*>(nshift + f); }
                    if (BF) {
                        typedef H bf16x4 __attribute__((ext_vector_type(4)));
                        bf16x4 pk = {static_cast<H>(v[0]), static_cast<H>(v[1]), static_cast<H>(v[2]), static_cast<H>(v[3])};
                        {   // 8-byte chunk c8 of row `row` lives at chunk c8 ^ (row & 31): the 16 rows x 4 chunks of one store land on 64 banks
                            const int row = 32 * gw + 16 * g + fr_e, c8 = f >> 2;
                            *reinterpret_cast<bf16x4*>(otile + row * 256 + ((c8 ^ (row & 31)) << 3)) = pk;
                        }
                        if (pre) {
                            bf16x4 pp;
#pragma unroll
                            for (int e = 0; e < 4; ++e) { const float t = fmaf(static_cast<float>(pk[e]), nsc[e], nsh[e]); pp[e] = static_cast<H>(t > 0.0f ? t : 0.3f * t); }
                            *reinterpret_cast<bf16x4*>(pre + oi) = pp;
                        }
                    } else {
                        *reinterpret_cast<f32x4*>(out + oi) = f32x4{v[0], v[1], v[2], v[3]};
                        if (pre) {
                            f32x4 pp;
#pragma unroll
                            for (int e = 0; e < 4; ++e) { const float t = fmaf(v[e], nsc[e], nsh[e]); pp[e] = t > 0.0f ? t : 0.3f * t; }
                            *reinterpret_cast<f32x4*>(pre + oi) = pp;
                        }
                    }
                }
            }
        }
        tp_prev = tp0; b_prev = b;
    }
    if (BF && tp_prev >= 0) {
        __syncthreads();
        flush(b_prev, tp_prev, smem + CF::OUT_OFF + ((item1 - 1 - item0) & 1) * CF::OUT_BYTES);
    }
}

// four consecutive channels c .. c + 3 (c % 4 == 0) of row `row` into the S32 split layout (per row, per 32 channels: 32 hi | 32 lo halves):
// the operand format of the split convolution kernels (r2_step.hip); C channels per row
__device__ __forceinline__ void store_s32_chunk(char* base, int64_t row, int C, int c, const float (&v)[4]) {
    x3x4_t hi, lo;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const x3_t hb = x3_hi(v[j]);
        hi[j] = hb;
        lo[j] = x3_lo(v[j], hb);
    }
    char* q = base + row * (int64_t)C * 4 + (c >> 5) * 128 + (c & 31) * 2;
    *reinterpret_cast<x3x4_t*>(q) = hi;
    *reinterpret_cast<x3x4_t*>(q + 64) = lo;
}

// ---- the sinc front-end of F32X3 handles: fp32 in / out, products as three fp16 MFMAs on half hi | lo parts (round 4) ----------------------
// Same structure as the 16-bit path of rn_sinc_kernel (LayerNorm output staged by 4-byte LDS-DMA into eight sample-shifted copies, filter
// fragments in registers, v_mfma_f32_16x16x32_f16, pooled tile through an LDS image), with a second plane for everything: the lo parts of the
// waveform (rn_ln_stats<LO>) and of the filters.  A wave owns 32 filters (hi and lo fragments: 128 VGPRs), all four waves the same 32 pooled
// frames of a tile; per (k step, pool partner) two fragment reads feed six MFMAs (lo.hi + hi.lo + hi.hi for two filter blocks).  fp32 output.
// The exact-fp32-MFMA instance of rn_sinc_kernel it replaces on these handles ran at the fp32 matrix peak: 3.7 ms per 256 utterances.
constexpr int SX_PT = 32;                   // pooled frames per tile
constexpr int SX_SAMPLES = 360;             // >= 3 * SX_PT conv positions + 255 + 8, multiple of 8
struct SincX3Cfg {
    static constexpr int COPY_BYTES = 1024;                    // 720 bytes of samples + up to 240 of skew
    static constexpr unsigned COPY_SKEW = 0x7E49C210u;         // (the fragment pattern of the 16-bit kernel: SincCfg16)
    static constexpr int PLANE = 8 * COPY_BYTES;
    static constexpr int XLDS = 2 * PLANE;                     // hi | lo planes of one operand buffer
    static constexpr int OUT_OFF = 2 * XLDS;                   // two operand buffers, then two output images
    static constexpr int OUT_BYTES = SX_PT * 512;              // 32 pooled frames x 128 filters fp32
    static constexpr int BN_OFF = OUT_OFF + 2 * OUT_BYTES;
    static constexpr int LDS = BN_OFF + 1024;                  // 65 KiB: two workgroups per CU
};

// filt: [2][128][256] half (hi plane, lo plane; k contiguous, zero beyond 251); xn: per utterance four rows of Lp halves (rn_ln_stats<LO>)
// pre32 (optional): block 0's pre-activation lrelu0.3(bn1(x)) in the S32 split layout, written from the same LDS image (the operand of the
// split convolution kernel: no separate rn_bn_act pass over the 1.4 GB front-end output)
__global__ __launch_bounds__(256, 2) void rn_sinc_x3_kernel(const f16_t* __restrict__ filt, const float* __restrict__ bn_scale, const float* __restrict__ bn_shift,
                                                            float* __restrict__ out, int T1, int B, const uint16_t* __restrict__ xn, int Lp,
                                                            char* __restrict__ pre32, const float* __restrict__ nscale, const float* __restrict__ nshift) {
    typedef SincX3Cfg CF;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tiles_u = (T1 + SX_PT - 1) / SX_PT;
    const int items = B * tiles_u;
    const int per = (items + (int)gridDim.x - 1) / (int)gridDim.x;
    const int item0 = (int)blockIdx.x * per, item1 = item0 + per < items ? item0 + per : items;
    if (item0 >= items) return;                                  // workgroup-uniform
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // = filter quarter: filters 32 wave .. + 31
    const int r16 = lane & 15, q4 = lane >> 4;

    bf16x8 whi[2][8], wlo[2][8];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        const char* w = reinterpret_cast<const char*>(filt) + (int64_t)(wave * 32 + blk * 16 + r16) * 256 * 2 + q4 * 16;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            whi[blk][kk] = *reinterpret_cast<const bf16x8*>(w + kk * 64);
            wlo[blk][kk] = *reinterpret_cast<const bf16x8*>(w + 128 * 256 * 2 + kk * 64);
        }
    }
    // (a thread's chunk column c16 = tid & 31 is the same in every trip of the flush: its four bn1 constants stay in registers)
    f32x4 nsc4 = {0.f, 0.f, 0.f, 0.f}, nsh4 = {0.f, 0.f, 0.f, 0.f};
    if (pre32) { nsc4 = *reinterpret_cast<const f32x4*>(nscale + (tid & 31) * 4); nsh4 = *reinterpret_cast<const f32x4*>(nshift + (tid & 31) * 4); }
    auto flush = [&](int b, int tpf, const char* img) {              // 32 rows x 32 chunks of 16 bytes, 4 per thread
        int tid_f = tid;
        asm volatile("" : "+v"(tid_f));
#pragma unroll
        for (int e = 0; e < SX_PT * 32 / 256; ++e) {
            const int idx = e * 256 + tid_f;
            const int row = idx >> 5, c16 = idx & 31;
            const f32x4 t = *reinterpret_cast<const f32x4*>(img + row * 512 + ((c16 ^ (row & 31)) << 4));
            if (tpf + row < T1) {
                const int64_t grow = (int64_t)b * T1 + tpf + row;
                *reinterpret_cast<f32x4*>(out + grow * 128 + c16 * 4) = t;
                if (pre32) {
                    float pv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { const float w = fmaf(t[u], nsc4[u], nsh4[u]); pv[u] = w > 0.0f ? w : 0.3f * w; }
                    store_s32_chunk(pre32, grow, 128, c16 * 4, pv);
                }
            }
        }
    };
    typedef __attribute__((address_space(3))) void lds_void_t;
    typedef __attribute__((address_space(1))) const void gbl_void_t;
    auto dma = [&](int item, int buf) {                          // plane pl, copy pc, index i <- part pl of sample 3 * tpn + i + pc of utterance b
        const int b = item / tiles_u, tpn = (item - b * tiles_u) * SX_PT;
        const uint16_t* src0 = xn + (int64_t)b * 4 * Lp + 3 * tpn + 2 * lane;
        char* dst = smem + buf * CF::XLDS;
#pragma unroll
        for (int e = 0; e < 12; ++e) {                               // 2 planes x 8 copies x 3 segments of 128 samples = 48 wave-instructions, 12 per wave
            const int idx = wave + 4 * e, pl = idx / 24, rem = idx - pl * 24, pc = rem / 3, q = rem - pc * 3;
            const uint16_t* src = src0 + pl * 2 * Lp + (pc & 1) * Lp + (pc & ~1) + q * 128;
            if (q * 128 + 2 * lane < SX_SAMPLES)
                __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(dst + pl * CF::PLANE + pc * CF::COPY_BYTES + ((CF::COPY_SKEW >> (4 * pc)) & 15) * 16 + q * 256), 4, 0, 0);
        }
    };
    float* const bnl = reinterpret_cast<float*>(smem + CF::BN_OFF);
    bnl[tid] = tid < 128 ? bn_scale[tid] : bn_shift[tid - 128];
    dma(item0, 0);
    int tp_prev = -1, b_prev = 0;
    for (int item = item0; item < item1; ++item) {
        const int it = item - item0;
        const int b = item / tiles_u;
        const int tp0 = (item - b * tiles_u) * SX_PT;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        char* const xbuf = smem + (it & 1) * CF::XLDS;
        char* const otile = smem + CF::OUT_OFF + (it & 1) * CF::OUT_BYTES;
        if (item + 1 < item1) dma(item + 1, (it + 1) & 1);
        if (tp_prev >= 0) flush(b_prev, tp_prev, smem + CF::OUT_OFF + ((it - 1) & 1) * CF::OUT_BYTES);
        const char* base[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int s = 3 * r16 + j;                                   // conv position inside the tile (frame group 0; group 1: + 48)
            base[j] = xbuf + (s & 7) * CF::COPY_BYTES + (((CF::COPY_SKEW >> (4 * (s & 7))) & 15) + (s >> 3) + q4) * 16;
        }
        int r16_e = r16, q4_e = q4;
        asm volatile("" : "+v"(r16_e), "+v"(q4_e));
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            f32x4 acc[2][3];
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[blk][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            // 24 (k step, pool partner) pairs; the hi and lo fragments of a pair are read two pairs ahead of their six MFMAs, fenced
            auto xread = [&](int idx, int pl) { return *reinterpret_cast<const bf16x8*>(base[idx % 3] + pl * CF::PLANE + g * 96 + (idx / 3) * 64); };
            bf16x8 rh[3], rl[3];
            rh[0] = xread(0, 0); rl[0] = xread(0, 1);
            rh[1] = xread(1, 0); rl[1] = xread(1, 1);
#pragma unroll
            for (int idx = 0; idx < 24; ++idx) {
                if (idx + 2 < 24) { rh[(idx + 2) % 3] = xread(idx + 2, 0); rl[(idx + 2) % 3] = xread(idx + 2, 1); }
                __builtin_amdgcn_sched_barrier(0);
                const int kk = idx / 3, j = idx % 3;
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) acc[blk][j] = Half16<f16_t>::mfma16(wlo[blk][kk], rh[idx % 3], acc[blk][j]);      // small terms first
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) acc[blk][j] = Half16<f16_t>::mfma16(whi[blk][kk], rl[idx % 3], acc[blk][j]);
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) acc[blk][j] = Half16<f16_t>::mfma16(whi[blk][kk], rh[idx % 3], acc[blk][j]);
                __builtin_amdgcn_sched_barrier(0);
            }
            // |.| -> max over the 3 pool partners -> BN -> LeakyReLU(0.3): lane = pooled frame 16 g + r16, filters 32 wave + 16 blk + 4 q4 .. + 3
            const int row = 16 * g + r16_e;
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const int f = wave * 32 + blk * 16 + 4 * q4_e;
                const f32x4 sc = *reinterpret_cast<const f32x4*>(bnl + f);
                const f32x4 sh = *reinterpret_cast<const f32x4*>(bnl + 128 + f);
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float m = fmaxf(fmaxf(fabsf(acc[blk][0][e]), fabsf(acc[blk][1][e])), fabsf(acc[blk][2][e]));
                    const float y = fmaf(m, sc[e], sh[e]);
                    v[e] = y > 0.0f ? y : 0.3f * y;
                }
                *reinterpret_cast<f32x4*>(otile + row * 512 + (((f >> 2) ^ (row & 31)) << 4)) = v;
            }
        }
        tp_prev = tp0; b_prev = b;
    }
    if (tp_prev >= 0) {
        __syncthreads();
        flush(b_prev, tp_prev, smem + CF::OUT_OFF + ((item1 - 1 - item0) & 1) * CF::OUT_BYTES);
    }
}

template <typename T, bool S32 = false>
__global__ __launch_bounds__(256) void rn_bn_act_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, int C, float slope, int64_t chunks) {
    constexpr int VEC = Vec16<T>::N;
    const int cpr = C / VEC;
    for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < chunks; id += (int64_t)gridDim.x * 256) {
        const int c = (int)(id % cpr) * VEC;
        Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(x + id * VEC), o;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float t = fmaf(v.get(j), scale[c + j], shift[c + j]);
            o.set(j, t > 0.0f ? t : slope * t);
        }
        if constexpr (S32) {           // (fp32 only: y is the S32 image of the result)
            const float vv[4] = {o.get(0), o.get(1), o.get(2), o.get(3)};
            store_s32_chunk(reinterpret_cast<char*>(y), id / cpr, C, c, vv);
        } else
        *reinterpret_cast<Vec16<T>*>(y + id * VEC) = o;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void rn_maxpool3_kernel(const T* __restrict__ x, T* __restrict__ y, int Tin, int Tout, int C,
                                                          int64_t chunks) {
    constexpr int VEC = Vec16<T>::N;
    const int cpr = C / VEC;
    for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < chunks; id += (int64_t)gridDim.x * 256) {
        const int c = (int)(id % cpr) * VEC;
        const int64_t row = id / cpr;                       // b * Tout + t
        const int64_t b = row / Tout;
        const int t = (int)(row - b * Tout);
        const T* p = x + ((b * Tin + 3 * t) * (int64_t)C + c);
        Vec16<T> a = *reinterpret_cast<const Vec16<T>*>(p), bb = *reinterpret_cast<const Vec16<T>*>(p + C),
                 cc = *reinterpret_cast<const Vec16<T>*>(p + 2 * C), o;
#pragma unroll
        for (int j = 0; j < VEC; ++j) o.set(j, fmaxf(fmaxf(a.get(j), bb.get(j)), cc.get(j)));
        *reinterpret_cast<Vec16<T>*>(y + row * C + c) = o;
    }
}

// AFMS gate; with `pre` the next block's pre-activation lrelu(bn1(.)) (or the aggregation BN) is written in the same pass,
// computed from the value as stored (rounded to T), so the result equals a separate rn_bn_act over y bit for bit.
template <typename T, bool S32 = false>
__global__ __launch_bounds__(256) void rn_afms_apply_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ alpha,
                                                            const float* __restrict__ s, int Tn, int C, int64_t chunks,
                                                            const float* __restrict__ nscale, const float* __restrict__ nshift,
                                                            T* __restrict__ pre, float slope) {
    constexpr int VEC = Vec16<T>::N;
    const int cpr = C / VEC;
    for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < chunks; id += (int64_t)gridDim.x * 256) {
        const int c = (int)(id % cpr) * VEC;
        const int64_t b = (id / cpr) / Tn;
        Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(x + id * VEC), o;
#pragma unroll
        for (int j = 0; j < VEC; ++j) o.set(j, (v.get(j) + alpha[c + j]) * s[b * C + c + j]);
        if (y) *reinterpret_cast<Vec16<T>*>(y + id * VEC) = o;          // (null: the next block has a projection shortcut and never reads x)
        if (pre) {
            Vec16<T> q;
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float t = fmaf(o.get(j), nscale[c + j], nshift[c + j]);
                q.set(j, t > 0.0f ? t : slope * t);
            }
            if constexpr (S32) {       // (fp32 only: pre is the S32 image of the next pre-activation)
                const float vv[4] = {q.get(0), q.get(1), q.get(2), q.get(3)};
                store_s32_chunk(reinterpret_cast<char*>(pre), id / cpr, C, c, vv);
            } else
            *reinterpret_cast<Vec16<T>*>(pre + id * VEC) = q;
        }
    }
}

// one thread per (b, c): softmax over the Tn (= 14) frames of the fp32 logits, weighted mean / std of x
template <typename T>
__global__ __launch_bounds__(256) void rn_attn_pool_kernel(const float* __restrict__ logits, const T* __restrict__ x, int Tn, int C,
                                                           float* __restrict__ out, int B) {
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= B * C) return;
    const int b = id / C, c = id - b * C;
    const float* lg = logits + (int64_t)b * Tn * C + c;
    const T* xp = x + (int64_t)b * Tn * C + c;
    float mx = -INFINITY;
    for (int t = 0; t < Tn; ++t) mx = fmaxf(mx, lg[(int64_t)t * C]);
    float se = 0.0f;
    for (int t = 0; t < Tn; ++t) se += expf(lg[(int64_t)t * C] - mx);
    float m = 0.0f, q = 0.0f;
    for (int t = 0; t < Tn; ++t) {
        const float w = expf(lg[(int64_t)t * C] - mx) / se;
        const float xv = to_f32<T>(xp[(int64_t)t * C]);
        m = fmaf(xv, w, m);
        q = fmaf(xv * xv, w, q);
    }
    out[(int64_t)b * 2 * C + c] = m;
    out[(int64_t)b * 2 * C + C + c] = sqrtf(fmaxf(q - m * m, 1e-5f));
}


// ---- fused block tail: [max_pool1d(3)] -> AFMS (column mean, fc, sigmoid, (y + alpha) * gate) -> next block's lrelu(bn(.)) ----
// RawNet_baseline.py:62-68,228-229.  One workgroup per utterance: the pooled activation stays in registers (at most TAIL_NCH
// 16-byte chunks per thread) between the column-mean pass and the gated write, so the block output is read from HBM once
// and the four separate passes (rn_maxpool3, rn_afms_mean, rn_afms_gate, rn_afms_apply) become one launch.
// Measured and not kept: two utterances per workgroup sharing one read of the fc weight (C = 512: 1 MiB per workgroup, the
// matrix-vector product is ~25 of the ~45 us of the late blocks) — 54 us instead of 47, half as many workgroups each twice as
// long; a per-workgroup rotation of the row order (-10 us: the 256 workgroups read the same lines at the same time) would make
// an utterance's gate depend on its position in the batch; a bf16 copy of the fc weight (-48 us over the six tails) moves the
// embeddings by up to 4 % of their scale against the separate passes on random weights (the gate scales a whole block output).
constexpr int TAIL_THREADS = 1024, TAIL_NCH = 16;

// RES: x is conv2's output WITHOUT the identity shortcut (the persistent conv-gather GEMM has no residual operand); the block input
// `res` (B, Tin, C) is added here, before the pool — the sum is formed in fp32 and rounded once, where the GEMM epilogue rounded the
// conv output and then the sum.
template <typename T, bool POOL, bool RES>
__global__ __launch_bounds__(TAIL_THREADS) void rn_tail_kernel(const T* __restrict__ x, T* __restrict__ y, T* __restrict__ pre,
                                                               const float* __restrict__ alpha, const float* __restrict__ WT,
                                                               const float* __restrict__ bias, const float* __restrict__ nscale,
                                                               const float* __restrict__ nshift, int Tin, int Tn, int C, float slope,
                                                               const T* __restrict__ res, int pre_s32) {
    constexpr int VEC = Vec16<T>::N;
    __shared__ float part[8192];                     // 32 KiB: column-sum partials, then the gate's partial dot products
    __shared__ float mean[512], gate[512];
    const int tid = threadIdx.x;
    const int64_t b = blockIdx.x;
    const int cpr = C / VEC, rstep = TAIL_THREADS / cpr;
    const int cc = tid % cpr, r0 = tid / cpr, c = cc * VEC;
    const T* xb = x + b * Tin * (int64_t)C + c;
    const T* rb = RES ? res + b * Tin * (int64_t)C + c : nullptr;
    Vec16<T> held[TAIL_NCH];
    float sum[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) sum[j] = 0.0f;
#pragma unroll
    for (int i = 0; i < TAIL_NCH; ++i) {
        const int t = r0 + i * rstep;
        if (t < Tn) {
            if (POOL) {
                const T* q = xb + (int64_t)(3 * t) * C;
                const Vec16<T> a = ld_nt<T>(q), bb = ld_nt<T>(q + C), d = ld_nt<T>(q + 2 * C);      // (conv2's output and the block input: last reads)
                if (RES) {
                    const T* r = rb + (int64_t)(3 * t) * C;
                    const Vec16<T> ra = ld_nt<T>(r), rbb = ld_nt<T>(r + C), rd = ld_nt<T>(r + 2 * C);
#pragma unroll
                    for (int j = 0; j < VEC; ++j) held[i].set(j, fmaxf(fmaxf(a.get(j) + ra.get(j), bb.get(j) + rbb.get(j)), d.get(j) + rd.get(j)));
                } else {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) held[i].set(j, fmaxf(fmaxf(a.get(j), bb.get(j)), d.get(j)));
                }
            } else {
                held[i] = ld_nt<T>(xb + (int64_t)t * C);
                if (RES) {
                    const Vec16<T> r = ld_nt<T>(rb + (int64_t)t * C);
#pragma unroll
                    for (int j = 0; j < VEC; ++j) held[i].set(j, held[i].get(j) + r.get(j));
                }
            }
#pragma unroll
            for (int j = 0; j < VEC; ++j) sum[j] += held[i].get(j);
        }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) part[r0 * C + c + j] = sum[j];
    __syncthreads();
    if (tid < C) {
        float m = 0.0f;
        for (int r = 0; r < rstep; ++r) m += part[r * C + tid];
        mean[tid] = m / (float)Tn;
    }
    __syncthreads();
    // gate[n] = sigmoid(bias[n] + sum_c WT[c][n] * mean[c]): thread = (4 channels n, one slice of c); WT is the fc weight transposed
    {
        const int n4 = C / 4, slices = TAIL_THREADS / n4, per = C / slices;
        const int nq = tid % n4, sl = tid / n4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* w = WT + (int64_t)(sl * per) * C + nq * 4;
#pragma unroll 8
        for (int k = 0; k < per; ++k) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(w + (int64_t)k * C);
            const float mv = mean[sl * per + k];
            acc[0] = fmaf(wv[0], mv, acc[0]); acc[1] = fmaf(wv[1], mv, acc[1]); acc[2] = fmaf(wv[2], mv, acc[2]); acc[3] = fmaf(wv[3], mv, acc[3]);
        }
        *reinterpret_cast<f32x4*>(&part[sl * C + nq * 4]) = acc;
        __syncthreads();
        if (tid < C) {
            float a = bias[tid];
            for (int q = 0; q < slices; ++q) a += part[q * C + tid];
            gate[tid] = 1.0f / (1.0f + expf(-a));
        }
        __syncthreads();
    }
    float al[VEC], g[VEC], ns[VEC], nh[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        al[j] = alpha[c + j]; g[j] = gate[c + j];
        ns[j] = pre ? nscale[c + j] : 0.0f; nh[j] = pre ? nshift[c + j] : 0.0f;
    }
    T* yb = y ? y + b * Tn * (int64_t)C + c : nullptr;     // (null: nothing reads the block output itself, only its pre-activation)
    T* pb = pre ? pre + b * Tn * (int64_t)C + c : nullptr;
#pragma unroll
    for (int i = 0; i < TAIL_NCH; ++i) {
        const int t = r0 + i * rstep;
        if (t < Tn) {
            Vec16<T> o, q;
#pragma unroll
            for (int j = 0; j < VEC; ++j) o.set(j, (held[i].get(j) + al[j]) * g[j]);
            if (yb) *reinterpret_cast<Vec16<T>*>(yb + (int64_t)t * C) = o;
            if (pb) {                                  // from the value as stored (rounded to T), like rn_afms_apply
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const float v = fmaf(o.get(j), ns[j], nh[j]);
                    q.set(j, v > 0.0f ? v : slope * v);
                }
                if constexpr (sizeof(T) == 4) {
                    if (pre_s32) {          // (fp32 handles: the next block's operand in the S32 split layout)
                        const float vv[4] = {q.get(0), q.get(1), q.get(2), q.get(3)};
                        store_s32_chunk(reinterpret_cast<char*>(pre), b * Tn + t, C, c, vv);
                    } else *reinterpret_cast<Vec16<T>*>(pb + (int64_t)t * C) = q;
                } else *reinterpret_cast<Vec16<T>*>(pb + (int64_t)t * C) = q;
            }
        }
    }
}

// ---- the same tail for SMALL batches (the reference API's per-file calls: B = 10 - 20 crops) ---------------------------------------
// One workgroup per utterance is 10 - 20 workgroups on 256 CUs, each walking load -> reduce -> matrix-vector -> store alone (33 us per
// launch, six launches: a quarter of a B = 20 call).  Here an utterance is cut into S frame slices: rn_tail_part (grid S x B) writes the
// slices' column sums of the pooled (+ residual) activation, rn_afms_gate (rn_block128.hip) turns them into the gate, rn_tail_apply
// (grid S x B) forms the same values again and writes (v + alpha) * gate and the next pre-activation.  Same arithmetic per element as
// rn_tail_kernel; the column mean is summed in a different order (fp32 round-off; a batch takes ONE of the two forms, by its size alone).
constexpr int TAILS_THREADS = 256, TAILS_MAX_S = 16;

template <typename T, bool POOL, bool RES>
__device__ __forceinline__ Vec16<T> tail_value(const T* xb, const T* rb, int t, int C) {
    constexpr int VEC = Vec16<T>::N;
    Vec16<T> v;
    if (POOL) {
        const T* q = xb + (int64_t)(3 * t) * C;
        const Vec16<T> a = ld_nt<T>(q), bb = ld_nt<T>(q + C), d = ld_nt<T>(q + 2 * C);
        if (RES) {
            const T* r = rb + (int64_t)(3 * t) * C;
            const Vec16<T> ra = ld_nt<T>(r), rbb = ld_nt<T>(r + C), rd = ld_nt<T>(r + 2 * C);
#pragma unroll
            for (int j = 0; j < VEC; ++j) v.set(j, fmaxf(fmaxf(a.get(j) + ra.get(j), bb.get(j) + rbb.get(j)), d.get(j) + rd.get(j)));
        } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j) v.set(j, fmaxf(fmaxf(a.get(j), bb.get(j)), d.get(j)));
        }
    } else {
        v = ld_nt<T>(xb + (int64_t)t * C);
        if (RES) {
            const Vec16<T> r = ld_nt<T>(rb + (int64_t)t * C);
#pragma unroll
            for (int j = 0; j < VEC; ++j) v.set(j, v.get(j) + r.get(j));
        }
    }
    return v;
}

// part[(b * S + s) * C + c] = sum over the frames of slice s of utterance b
template <typename T, bool POOL, bool RES>
__global__ __launch_bounds__(TAILS_THREADS) void rn_tail_part_kernel(const T* __restrict__ x, const T* __restrict__ res, float* __restrict__ part,
                                                                     int Tin, int Tn, int C, int per) {
    constexpr int VEC = Vec16<T>::N;
    __shared__ float red[TAILS_THREADS / 32][512];     // [row group][channel]: cpr >= 32 (C >= 256 at VEC 8, >= 128 at VEC 4) -> <= 8 row groups
    const int tid = threadIdx.x, s = blockIdx.x, S = gridDim.x;
    const int64_t b = blockIdx.y;
    const int cpr = C / VEC, rstep = TAILS_THREADS / cpr;
    const int cc = tid % cpr, r0 = tid / cpr, c = cc * VEC;
    const T* xb = x + b * Tin * (int64_t)C + c;
    const T* rb = RES ? res + b * Tin * (int64_t)C + c : nullptr;
    const int t0 = s * per, t1 = min(Tn, t0 + per);
    float sum[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) sum[j] = 0.0f;
#pragma unroll 4
    for (int t = t0 + r0; t < t1; t += rstep) {
        const Vec16<T> v = tail_value<T, POOL, RES>(xb, rb, t, C);
#pragma unroll
        for (int j = 0; j < VEC; ++j) sum[j] += v.get(j);
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) red[r0][c + j] = sum[j];
    __syncthreads();
    for (int n = tid; n < C; n += TAILS_THREADS) {
        float a = 0.0f;
        for (int r = 0; r < rstep; ++r) a += red[r][n];
        part[(b * S + s) * C + n] = a;
    }
}

template <typename T, bool POOL, bool RES>
__global__ __launch_bounds__(TAILS_THREADS) void rn_tail_apply_kernel(const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y, T* __restrict__ pre,
                                                                      const float* __restrict__ alpha, const float* __restrict__ gate,
                                                                      const float* __restrict__ nscale, const float* __restrict__ nshift,
                                                                      int Tin, int Tn, int C, int per, float slope, int pre_s32) {
    constexpr int VEC = Vec16<T>::N;
    const int tid = threadIdx.x, s = blockIdx.x;
    const int64_t b = blockIdx.y;
    const int cpr = C / VEC, rstep = TAILS_THREADS / cpr;
    const int cc = tid % cpr, r0 = tid / cpr, c = cc * VEC;
    const T* xb = x + b * Tin * (int64_t)C + c;
    const T* rb = RES ? res + b * Tin * (int64_t)C + c : nullptr;
    float al[VEC], g[VEC], ns[VEC], nh[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        al[j] = alpha[c + j]; g[j] = gate[b * C + c + j];
        ns[j] = pre ? nscale[c + j] : 0.0f; nh[j] = pre ? nshift[c + j] : 0.0f;
    }
    T* yb = y ? y + b * Tn * (int64_t)C + c : nullptr;
    T* pb = pre ? pre + b * Tn * (int64_t)C + c : nullptr;
    const int t0 = s * per, t1 = min(Tn, t0 + per);
#pragma unroll 4
    for (int t = t0 + r0; t < t1; t += rstep) {
        const Vec16<T> v = tail_value<T, POOL, RES>(xb, rb, t, C);
        Vec16<T> o, q;
#pragma unroll
        for (int j = 0; j < VEC; ++j) o.set(j, (v.get(j) + al[j]) * g[j]);
        if (yb) *reinterpret_cast<Vec16<T>*>(yb + (int64_t)t * C) = o;
        if (pb) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float w = fmaf(o.get(j), ns[j], nh[j]);
                q.set(j, w > 0.0f ? w : slope * w);
            }
            if constexpr (sizeof(T) == 4) {
                    if (pre_s32) {          // (fp32 handles: the next block's operand in the S32 split layout)
                        const float vv[4] = {q.get(0), q.get(1), q.get(2), q.get(3)};
                        store_s32_chunk(reinterpret_cast<char*>(pre), b * Tn + t, C, c, vv);
                    } else *reinterpret_cast<Vec16<T>*>(pb + (int64_t)t * C) = q;
                } else *reinterpret_cast<Vec16<T>*>(pb + (int64_t)t * C) = q;
        }
    }
}

inline int grid_for(int64_t items) {
    int64_t g = (items + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace

hipError_t launch_rn_ln_stats(const float* wav, int B, int L, float* stats, hipStream_t stream, void* xn, int Lp, const float* gamma,
                              const float* beta, int xn_dt, bool xn_lo) {
    if (xn && (Lp < L + RN_XN_TAIL || Lp % 64 != 0 || !gamma || !beta)) return hipErrorInvalidValue;
    // the register-resident form: whole 8-sample units, 16-byte aligned rows and vectors, the padded row within 8 192 NU samples
    const int span = xn ? Lp : L;
    const bool reg_ok = L % 8 == 0 && L >= 8 && ((reinterpret_cast<uintptr_t>(wav) | reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta) |
                                                   reinterpret_cast<uintptr_t>(xn)) & 15) == 0 && (!xn || Lp % 8 == 0) && span <= 8192 * 8;
    if (reg_ok) {
#define SV_LN(HH, LOO)                                                                                                                        \
        {                                                                                                                                     \
            if (span <= 8192 * 4) hipLaunchKernelGGL((rn_ln_stats_reg_kernel<HH, LOO, 4>), dim3(B), dim3(1024), 0, stream, wav, L, stats, reinterpret_cast<HH*>(xn), Lp, gamma, beta); \
            else hipLaunchKernelGGL((rn_ln_stats_reg_kernel<HH, LOO, 8>), dim3(B), dim3(1024), 0, stream, wav, L, stats, reinterpret_cast<HH*>(xn), Lp, gamma, beta);            \
            return hipGetLastError();                                                                                                         \
        }
        if (xn_lo) SV_LN(f16_t, true)
        if (xn_dt == DT_F16) SV_LN(f16_t, false)
        SV_LN(bf16_t, false)
#undef SV_LN
    }
    if (xn_lo) hipLaunchKernelGGL((rn_ln_stats_kernel<f16_t, true>), dim3(B), dim3(256), 0, stream, wav, L, stats, reinterpret_cast<f16_t*>(xn), Lp, gamma, beta);
    else if (xn_dt == DT_F16) hipLaunchKernelGGL(rn_ln_stats_kernel<f16_t>, dim3(B), dim3(256), 0, stream, wav, L, stats, reinterpret_cast<f16_t*>(xn), Lp, gamma, beta);
    else hipLaunchKernelGGL(rn_ln_stats_kernel<bf16_t>, dim3(B), dim3(256), 0, stream, wav, L, stats, reinterpret_cast<bf16_t*>(xn), Lp, gamma, beta);
    return hipGetLastError();
}

hipError_t launch_rn_sinc(const float* wav, const float* stats, const float* gamma, const float* beta, const void* filt,
                          const float* bn_scale, const float* bn_shift, void* out, int dt, int B, int L, int T1,
                          hipStream_t stream, void* pre, const float* next_scale, const float* next_shift, const void* xn, int Lp, int num_cu, bool sym) {
    const bool bf16 = dt != DT_F32;
    if (sym && dt != DT_F16) return hipErrorInvalidValue;             // (`filt` is then the [128][128] slot-major table of the symmetric form)
    if (B <= 0) return hipErrorInvalidValue;
    if (T1 != (L - 250) / 3 || L < 251 + 3 || (pre && (!next_scale || !next_shift))) return hipErrorInvalidValue;
    if (bf16 && (!xn || Lp < L + RN_XN_TAIL || Lp % 64 != 0)) return hipErrorInvalidValue;          // the bf16 kernel stages from the normalised copy
    const int64_t items = (int64_t)B * ((T1 + SINC_PT - 1) / SINC_PT);
    if (items >= (1ll << 31)) return hipErrorInvalidValue;
    const int slots = 2 * (num_cu > 0 ? num_cu : 256);               // two workgroups per CU (256 VGPRs per lane, 47 KiB of LDS each)
    dim3 grid((unsigned)(items < slots ? items : slots)), block(256);
    if (dt == DT_F16 && sym)
        hipLaunchKernelGGL((rn_sinc_kernel<f16_t, true>), grid, block, SincCfg<f16_t>::LDS, stream, wav, stats, gamma, beta, filt, bn_scale,
                           bn_shift, reinterpret_cast<f16_t*>(out), reinterpret_cast<f16_t*>(pre), next_scale, next_shift, L, T1, B,
                           reinterpret_cast<const uint16_t*>(xn), Lp);
    else if (dt == DT_F16)
        hipLaunchKernelGGL(rn_sinc_kernel<f16_t>, grid, block, SincCfg<f16_t>::LDS, stream, wav, stats, gamma, beta, filt, bn_scale,
                           bn_shift, reinterpret_cast<f16_t*>(out), reinterpret_cast<f16_t*>(pre), next_scale, next_shift, L, T1, B,
                           reinterpret_cast<const uint16_t*>(xn), Lp);
    else if (bf16)
        hipLaunchKernelGGL(rn_sinc_kernel<bf16_t>, grid, block, SincCfg<bf16_t>::LDS, stream, wav, stats, gamma, beta, filt, bn_scale,
                           bn_shift, reinterpret_cast<bf16_t*>(out), reinterpret_cast<bf16_t*>(pre), next_scale, next_shift, L, T1, B,
                           reinterpret_cast<const uint16_t*>(xn), Lp);
    else
        hipLaunchKernelGGL(rn_sinc_kernel<float>, grid, block, SincCfg<float>::LDS, stream, wav, stats, gamma, beta, filt, bn_scale,
                           bn_shift, reinterpret_cast<float*>(out), reinterpret_cast<float*>(pre), next_scale, next_shift, L, T1, B,
                           static_cast<const uint16_t*>(nullptr), 0);
    return hipGetLastError();
}

// F32X3 handles: out (B, T1, 128) fp32 from the split LayerNorm output (rn_ln_stats with xn_lo: four rows of Lp halves per utterance)
hipError_t launch_rn_sinc_x3(const void* filt_planes, const float* bn_scale, const float* bn_shift, float* out, int B, int L, int T1,
                             const void* xn, int Lp, int num_cu, hipStream_t stream, void* pre_s32, const float* next_scale, const float* next_shift) {
    if (pre_s32 && (!next_scale || !next_shift || ((reinterpret_cast<uintptr_t>(pre_s32) | reinterpret_cast<uintptr_t>(next_scale) | reinterpret_cast<uintptr_t>(next_shift)) & 15)))
        return hipErrorInvalidValue;
    if (B <= 0 || !filt_planes || !bn_scale || !bn_shift || !out || !xn) return hipErrorInvalidValue;
    if (T1 != (L - 250) / 3 || L < 251 + 3 || Lp < L + RN_XN_TAIL || Lp % 64 != 0) return hipErrorInvalidValue;
    if ((reinterpret_cast<uintptr_t>(filt_planes) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(xn)) & 15) return hipErrorInvalidValue;
    const int64_t items = (int64_t)B * ((T1 + SX_PT - 1) / SX_PT);
    if (items >= (1ll << 31)) return hipErrorInvalidValue;
    static DeviceOnce attr;
    if (hipError_t e = set_max_dynamic_lds(attr, reinterpret_cast<const void*>(rn_sinc_x3_kernel), SincX3Cfg::LDS)) return e;
    const int slots = 2 * (num_cu > 0 ? num_cu : 256);
    hipLaunchKernelGGL(rn_sinc_x3_kernel, dim3((unsigned)(items < slots ? items : slots)), dim3(256), SincX3Cfg::LDS, stream,
                       reinterpret_cast<const f16_t*>(filt_planes), bn_scale, bn_shift, out, T1, B, reinterpret_cast<const uint16_t*>(xn), Lp,
                       reinterpret_cast<char*>(pre_s32), next_scale, next_shift);
    return hipGetLastError();
}

hipError_t launch_rn_bn_act(const void* x, void* y, int dt, const float* scale, const float* shift, int64_t rows, int C,
                            float slope, hipStream_t stream, bool y_s32) {
    if (y_s32 && (dt != DT_F32 || C % 32 != 0 || (reinterpret_cast<uintptr_t>(y) & 127))) return hipErrorInvalidValue;
    const bool bf16 = dt == DT_BF16;
    const int vec = dt != DT_F32 ? 8 : 4;
    if (C % vec) return hipErrorInvalidValue;
    const int64_t chunks = rows * (C / vec);
    if (dt == DT_F16) hipLaunchKernelGGL(rn_bn_act_kernel<f16_t>, dim3(grid_for(chunks)), dim3(256), 0, stream, (const f16_t*)x, (f16_t*)y, scale, shift, C, slope, chunks);
    else if (bf16) hipLaunchKernelGGL(rn_bn_act_kernel<bf16_t>, dim3(grid_for(chunks)), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)y, scale, shift, C, slope, chunks);
    else if (y_s32) hipLaunchKernelGGL((rn_bn_act_kernel<float, true>), dim3(grid_for(chunks)), dim3(256), 0, stream, (const float*)x, (float*)y, scale, shift, C, slope, chunks);
    else hipLaunchKernelGGL(rn_bn_act_kernel<float>, dim3(grid_for(chunks)), dim3(256), 0, stream, (const float*)x, (float*)y, scale, shift, C, slope, chunks);
    return hipGetLastError();
}

hipError_t launch_rn_maxpool3(const void* x, void* y, int dt, int B, int Tin, int C, hipStream_t stream) {
    const bool bf16 = dt == DT_BF16;
    const int vec = dt != DT_F32 ? 8 : 4;
    if (C % vec) return hipErrorInvalidValue;
    const int Tout = Tin / 3;
    const int64_t chunks = (int64_t)B * Tout * (C / vec);
    if (dt == DT_F16) hipLaunchKernelGGL(rn_maxpool3_kernel<f16_t>, dim3(grid_for(chunks)), dim3(256), 0, stream, (const f16_t*)x, (f16_t*)y, Tin, Tout, C, chunks);
    else if (bf16) hipLaunchKernelGGL(rn_maxpool3_kernel<bf16_t>, dim3(grid_for(chunks)), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)y, Tin, Tout, C, chunks);
    else hipLaunchKernelGGL(rn_maxpool3_kernel<float>, dim3(grid_for(chunks)), dim3(256), 0, stream, (const float*)x, (float*)y, Tin, Tout, C, chunks);
    return hipGetLastError();
}

hipError_t launch_rn_afms_apply(const void* x, void* y, int dt, const float* alpha, const float* s, int B, int T, int C,
                                hipStream_t stream, const float* next_scale, const float* next_shift, void* pre, float slope, bool pre_s32) {
    if (pre_s32 && (dt != DT_F32 || !pre || C % 32 != 0 || (reinterpret_cast<uintptr_t>(pre) & 127))) return hipErrorInvalidValue;
    const bool bf16 = dt == DT_BF16;
    const int vec = dt != DT_F32 ? 8 : 4;
    if (C % vec || (pre && (!next_scale || !next_shift))) return hipErrorInvalidValue;
    const int64_t chunks = (int64_t)B * T * (C / vec);
    if (dt == DT_F16) hipLaunchKernelGGL(rn_afms_apply_kernel<f16_t>, dim3(grid_for(chunks)), dim3(256), 0, stream, (const f16_t*)x, (f16_t*)y, alpha, s, T, C, chunks, next_scale, next_shift, (f16_t*)pre, slope);
    else if (bf16) hipLaunchKernelGGL(rn_afms_apply_kernel<bf16_t>, dim3(grid_for(chunks)), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)y, alpha, s, T, C, chunks, next_scale, next_shift, (bf16_t*)pre, slope);
    else if (pre_s32) hipLaunchKernelGGL((rn_afms_apply_kernel<float, true>), dim3(grid_for(chunks)), dim3(256), 0, stream, (const float*)x, (float*)y, alpha, s, T, C, chunks, next_scale, next_shift, (float*)pre, slope);
    else hipLaunchKernelGGL(rn_afms_apply_kernel<float>, dim3(grid_for(chunks)), dim3(256), 0, stream, (const float*)x, (float*)y, alpha, s, T, C, chunks, next_scale, next_shift, (float*)pre, slope);
    return hipGetLastError();
}

bool rn_tail_supported(int dt, int Tn, int C) {
    const int vec = dt != DT_F32 ? 8 : 4;
    if (C % vec || C % 4 || C > 512 || Tn <= 0) return false;
    const int cpr = C / vec, n4 = C / 4;
    if (TAIL_THREADS % cpr || TAIL_THREADS % n4 || n4 > TAIL_THREADS || C % (TAIL_THREADS / n4)) return false;
    if ((TAIL_THREADS / cpr) * C > 8192 || (TAIL_THREADS / n4) * C > 8192) return false;      // LDS partials
    return (int64_t)Tn * cpr <= (int64_t)TAIL_NCH * TAIL_THREADS;                                // the utterance fits the registers
}

// slices per utterance of the small-batch form (0: one workgroup per utterance, rn_tail_kernel)
int rn_tail_slices(int dt, int B, int Tn, int C, int num_cu) {
    const int vec = dt != DT_F32 ? 8 : 4;
    if (C % vec || C > 512 || C / vec < 32 || TAILS_THREADS % (C / vec)) return 0;
    if (B * 4 > num_cu) {
        // full batches: only the short utterances of the wide late blocks, where one workgroup per utterance streams 1 MiB of gate weights
        // through its CU for a few KB of activations (16 frames per slice; the gate on the fp32 MFMA, rn_afms_gate_mfma)
        if (C < 256 || Tn > 48) return 0;
        return (Tn + 15) / 16;
    }
    // small batches: WITHIN this branch the slice count depends on the utterance alone (about 48 frames per slice).  Which branch a call
    // takes does depend on its batch (B * 4 > CUs above; launch_rn_afms_gate and launch_rowvec_linear switch kernels at B = 64 as well,
    // and with option `lanes` it is the lane's share of B that counts): an utterance's column means are then summed in another order, and its
    // embedding moves by fp32 round-off with the batch it rides in — measured at the B = 64 | 65 boundary: 5.5e-6 of the embedding scale
    // on f32x3 handles, 1.1e-5 on f32 (tests/test_gpu_rawnet2.py::test_an_utterance_embeds_alike_on_both_sides_of_the_batch_size_switches).
    // INTEGRATION.md states the bound.
    int S = (Tn + 47) / 48;
    if (S > TAILS_MAX_S) S = TAILS_MAX_S;
    return S >= 2 ? S : 0;
}

hipError_t launch_rn_tail(const void* x, void* y, void* pre, int dt, bool pool, const float* alpha, const float* WT, const float* bias,
                          const float* next_scale, const float* next_shift, int B, int Tin, int C, float slope, hipStream_t stream, const void* res,
                          float* part, float* gate, int num_cu, bool pre_s32) {
    if (pre_s32 && (dt != DT_F32 || !pre || C % 32 != 0 || (reinterpret_cast<uintptr_t>(pre) & 127))) return hipErrorInvalidValue;
    const int Tn = pool ? Tin / 3 : Tin;
    if (!x || (!y && !pre) || !alpha || !WT || !bias || B <= 0 || !rn_tail_supported(dt, Tn, C) || (pre && (!next_scale || !next_shift)))
        return hipErrorInvalidValue;
    if (res && dt == DT_F32) return hipErrorInvalidValue;       // (the residual form exists for the 16-bit handles, whose conv2 may run without one)
    const int S = (part && gate) ? rn_tail_slices(dt, B, Tn, C, num_cu) : 0;
    if (S > 0) {        // small batch: slice sums -> gate -> apply (part: B x S x C floats, gate: B x C floats)
        const int per = (Tn + S - 1) / S;
        const dim3 grid(S, B), block(TAILS_THREADS);
#define SV_PART(TT, P, R) hipLaunchKernelGGL((rn_tail_part_kernel<TT, P, R>), grid, block, 0, stream, (const TT*)x, (const TT*)res, part, Tin, Tn, C, per)
#define SV_APPLY(TT, P, R) hipLaunchKernelGGL((rn_tail_apply_kernel<TT, P, R>), grid, block, 0, stream, (const TT*)x, (const TT*)res, (TT*)y, (TT*)pre, \
                                              alpha, gate, next_scale, next_shift, Tin, Tn, C, per, slope, pre_s32 ? 1 : 0)
#define SV_BOTH16(WHAT, TT) { if (res) { if (pool) WHAT(TT, true, true); else WHAT(TT, false, true); } else { if (pool) WHAT(TT, true, false); else WHAT(TT, false, false); } }
        if (dt == DT_F16) SV_BOTH16(SV_PART, f16_t)
        else if (dt == DT_BF16) SV_BOTH16(SV_PART, bf16_t)
        else { if (pool) SV_PART(float, true, false); else SV_PART(float, false, false); }
        if (hipError_t e = hipGetLastError()) return e;
        if (hipError_t e = launch_rn_afms_gate(part, S, B, C, Tn, WT, bias, gate, stream)) return e;
        if (dt == DT_F16) SV_BOTH16(SV_APPLY, f16_t)
        else if (dt == DT_BF16) SV_BOTH16(SV_APPLY, bf16_t)
        else { if (pool) SV_APPLY(float, true, false); else SV_APPLY(float, false, false); }
#undef SV_BOTH16
#undef SV_APPLY
#undef SV_PART
        return hipGetLastError();
    }
#define SV_TAIL(TT, P, R) hipLaunchKernelGGL((rn_tail_kernel<TT, P, R>), dim3(B), dim3(TAIL_THREADS), 0, stream, (const TT*)x, (TT*)y, (TT*)pre, \
                                             alpha, WT, bias, next_scale, next_shift, Tin, Tn, C, slope, (const TT*)res, pre_s32 ? 1 : 0)
#define SV_TAIL16(TT) { if (res) { if (pool) SV_TAIL(TT, true, true); else SV_TAIL(TT, false, true); } else { if (pool) SV_TAIL(TT, true, false); else SV_TAIL(TT, false, false); } }
    if (dt == DT_F16) SV_TAIL16(f16_t)
    else if (dt == DT_BF16) SV_TAIL16(bf16_t)
    else { if (pool) SV_TAIL(float, true, false); else SV_TAIL(float, false, false); }
#undef SV_TAIL16
#undef SV_TAIL
    return hipGetLastError();
}

hipError_t launch_rn_attn_pool(const float* logits, const void* x, int dt, int B, int T, int C, float* out, hipStream_t stream) {
    const int n = B * C;
    if (dt == DT_F16) hipLaunchKernelGGL(rn_attn_pool_kernel<f16_t>, dim3((n + 255) / 256), dim3(256), 0, stream, logits, (const f16_t*)x, T, C, out, B);
    else if (dt == DT_BF16) hipLaunchKernelGGL(rn_attn_pool_kernel<bf16_t>, dim3((n + 255) / 256), dim3(256), 0, stream, logits, (const bf16_t*)x, T, C, out, B);
    else hipLaunchKernelGGL(rn_attn_pool_kernel<float>, dim3((n + 255) / 256), dim3(256), 0, stream, logits, (const float*)x, T, C, out, B);
    return hipGetLastError();
}

}  // namespace svhip
