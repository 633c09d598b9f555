// res2net.hip — the whole Res2Net chain of one SE-Res2Net block in ONE launch (bf16 path).
//
// Reference: Res2NetBlock.forward, models/ECAPA_TDNN.py:118-129:  chunks c_0..c_7 of the tdnn1 output,
//   y_0 = c_0,  y_1 = B_0(c_1),  y_j = B_{j-1}(c_j + y_{j-1}),  B = BN(ReLU(conv_k3_dilated(.)))
// Seven dependent convolutions with only C/8 channels each: as separate GEMM launches they are 21
// latency-bound kernels per forward.  Here one workgroup owns one utterance:
//   * U = the current conv input (T x C/8, bf16) lives in LDS for the whole chain; the dilated taps
//     with reflect padding are just row-address arithmetic on it (no halo, no im2col);
//   * the weights of one tap (C/8 x C/8) are staged in LDS, the next tap's slab is prefetched into
//     registers under the MFMAs (global -> VGPR -> LDS, written after the barrier);
//   * 8 waves: n-tiles of 32 channels x interleaved 32-frame m-tiles, v_mfma_f32_32x32x16_bf16 with
//     the weights as the A operand, so a lane owns 4 consecutive output channels of one frame and
//     the epilogue (bias, ReLU, BatchNorm affine) writes packed bf16x4 straight back into U;
//   * a cooperative pass then streams y_j to HBM as whole rows and adds the next chunk c_{j+1}
//     in place (fp32 add, one bf16 rounding — same as the unfused path); the c_{j+1} rows are requested
//     before the last tap's MFMAs.
// Where the time goes (tools/res2_bench ablations, C = 1024, 150 us per launch): global traffic of the row passes 35 - 45 us,
// MFMAs 30 (28 at peak), epilogue 20, weight staging 12, barriers / loop skeleton 24.  All 256 workgroups reach their row
// passes together, so HBM sees 7 bursts of 60 MB per launch; measured and of no effect on the total: software-pipelined
// fragment reads, the c_{j+1} prefetch above, y_j stores delayed behind the next stage's weight loads (C = 512: 64 us either
// way).  What would help is two workgroups per CU out of phase, which the 136 KiB of LDS (C = 1024) rule out.  Round 3: the odd
// workgroups started 8 / 16 / 24 k cycles late (s_sleep) so that half of the CUs hit their row passes out of phase with the other
// half: 0.424 / 0.444 / 0.449 ms per three launches against 0.427 — the bursts are not what the time goes to; the same chain on 16 waves
// per workgroup (4 m-tiles per wave, 128 VGPRs): 0.46 ms — each weight fragment then feeds 4 MFMAs instead of 7 and the kernel spills.
//
// Time slices (round 4, small batches): one workgroup per utterance leaves most of the chip idle at the reference API's own batch size
// (B = num_eval = 10 - 20 crops of one file: 20 workgroups, ~96 us per launch whatever B is — the chain is seven dependent stages).
// With Res2Params::slices = S > 1 a workgroup owns the frames [c0, c1) of an utterance plus a halo of 7 * dil frames either side (the
// dependency cone of seven k = 3 stages): stage j leaves the outermost j * dil halo rows wrong, never a core row; only core rows are
// stored.  Same arithmetic per row, so the outputs are bit-identical to the whole-utterance form; fewer m-tiles per wave (MIT 3 / 2
// instead of 7 / 4), 80 / 48 KiB of LDS.  The launcher slices when S * B workgroups still fit the chip at once.
#include "common.h"
#include "kernels.h"

namespace svhip {

namespace {

#ifdef SVHIP_GEMM_DEBUG
constexpr bool R2DBG = true;         // ablations: 1 no global traffic in the row passes, 2 no MFMA, 4 no weight staging, 8 no epilogue
#else
constexpr bool R2DBG = false;
#endif
constexpr int R2_TMAX = 416;         // 13 MFMA m-tiles; T = 401 for 2 s @ 16 kHz

// MIT: m-tiles (32 frames) per wave: 7 / 4 = whole utterances of up to 416 frames, 3 / 2 = time slices of up to 192 / 256 rows
template <int CW, int MIT> struct R2Cfg {
    static constexpr int ROWB = CW * 2;                 // bytes per LDS row (256 / 128)
    static constexpr int NCH = ROWB / 16;               // 16-byte chunks per row (16 / 8)
    static constexpr int NT = CW / 32;                  // n-tiles (4 / 2)
    static constexpr int MW = 8 / NT;                   // wave groups along M (2 / 4)
    static constexpr int MI = MIT;                      // m-tiles per wave
    static constexpr int TMAX = MW * MI * 32 < R2_TMAX ? MW * MI * 32 : R2_TMAX;      // rows of the LDS image
    static constexpr int U_BYTES = TMAX * ROWB;
    static constexpr int W_BYTES = CW * ROWB;           // one tap: CW rows x CW k
    static constexpr int LDS = U_BYTES + W_BYTES;       // whole utterances: 136 KiB / 60 KiB; slices: 80 / 48 KiB
    static constexpr int WCH = CW * NCH / 512;          // weight chunks per thread per tap (4 / 1)
    static __device__ __forceinline__ int swz(int row) { return CW == 128 ? (row & 15) : ((row >> 1) & 7); }
};

template <int CW, int MIT>
__global__ __launch_bounds__(512, 2) void res2net_chain_kernel(Res2Params p) {
    typedef R2Cfg<CW, MIT> CF;
    constexpr int ROWB = CF::ROWB, NCH = CF::NCH, MI = CF::MI, WCH = CF::WCH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* U = smem;
    char* Wt = smem + CF::U_BYTES;

    // the rows this workgroup holds: local row r <-> frame l0 + r of utterance b; it stores the core frames [c0, c1)
    const int S = p.slices > 1 ? p.slices : 1;
    const int b = (int)blockIdx.x / S, si = (int)blockIdx.x - b * S;
    const int Tg = p.T;                                 // frames of the utterance (reflect padding acts at ITS ends)
    const int c0 = S > 1 ? si * p.Tc : 0, c1 = S > 1 ? min(Tg, c0 + p.Tc) : Tg;
    const int l0 = S > 1 ? max(0, c0 - 7 * p.dil) : 0, l1 = S > 1 ? min(Tg, c1 + 7 * p.dil) : Tg;
    const int T = l1 - l0;                              // local rows
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % CF::NT, wq = wave / CF::NT;
    const int fr = lane & 31, fh = lane >> 5;
    const bf16_t* __restrict__ H1 = reinterpret_cast<const bf16_t*>(p.H1) + ((int64_t)b * Tg + l0) * p.ld;
    bf16_t* __restrict__ H2 = reinterpret_cast<bf16_t*>(p.H2) + ((int64_t)b * Tg + l0) * p.ld;
    const int s0 = c0 - l0, s1 = c1 - l0;               // local rows that are stored

    // ---- weight slab prefetch (global -> registers), write (registers -> LDS) ----------------------
    u32x4 wregs[WCH];
    auto wload = [&](int layer, int tap) {
        const char* wsrc = reinterpret_cast<const char*>(p.W[layer]);
        int tid_w = tid;
        asm volatile("" : "+v"(tid_w));               // (addresses recomputed per call: hoisted, they spill)
#pragma unroll
        for (int e = 0; e < WCH; ++e) {
            const int c = tid_w + 512 * e;
            const int n = c / NCH, ch = c % NCH;
            wregs[e] = *reinterpret_cast<const u32x4*>(wsrc + ((int64_t)n * p.Kp + tap * CW) * 2 + ch * 16);
        }
    };
    auto wstore = [&]() {
        int tid_w = tid;
        asm volatile("" : "+v"(tid_w));
#pragma unroll
        for (int e = 0; e < WCH; ++e) {
            const int c = tid_w + 512 * e;
            const int n = c / NCH, ch = c % NCH;
            *reinterpret_cast<u32x4*>(Wt + n * ROWB + ((ch ^ CF::swz(n)) << 4)) = wregs[e];
        }
    };

    // ---- init: U = c_1, y_0 = c_0 straight to H2, first weight slab ---------------------------------
    wload(0, 0);
    {   // (loads of several row chunks in flight per thread, like the row passes below)
        constexpr int RSTEP = 512 / NCH, RB0 = 4;
        const int row0 = tid / NCH, ch0 = tid % NCH;
        const uint32_t goff0 = ((uint32_t)row0 * (uint32_t)p.ld + (uint32_t)ch0 * 8u) * 2u;
        const uint32_t gstep = (uint32_t)RSTEP * (uint32_t)p.ld * 2u;
        const char* h1b = reinterpret_cast<const char*>(H1);
        char* h2b = reinterpret_cast<char*>(H2);
        for (int it0 = 0; it0 * RSTEP < T; it0 += RB0) {
            u32x4 cc0[RB0], cc1[RB0];
#pragma unroll
            for (int k = 0; k < RB0; ++k)
                if (row0 + (it0 + k) * RSTEP < T) {
                    cc0[k] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(h1b + goff0 + (uint32_t)(it0 + k) * gstep));      // (H1: every line is read once)
                    cc1[k] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(h1b + CW * 2 + goff0 + (uint32_t)(it0 + k) * gstep));
                }
#pragma unroll
            for (int k = 0; k < RB0; ++k) {
                const int row = row0 + (it0 + k) * RSTEP;
                if (row < T) {
                    if (row >= s0 && row < s1) *reinterpret_cast<u32x4*>(h2b + goff0 + (uint32_t)(it0 + k) * gstep) = cc0[k];
                    *reinterpret_cast<u32x4*>(U + row * ROWB + ((ch0 ^ CF::swz(row)) << 4)) = cc1[k];
                }
            }
        }
    }
    wstore();
    __syncthreads();
    // debug builds, bit 64: wave 0 stamps the end of each phase of each stage (s_memtime; the phases end in barriers, so its clock is the workgroup's)
    const bool stamp = R2DBG && (p.debug & 64) && p.ts && tid == 0;
    unsigned long long tprev = 0;
    if (R2DBG && (p.debug & 64) && p.ts) tprev = __builtin_readcyclecounter();
    if (stamp) p.ts[(size_t)blockIdx.x * 32 + 3] = tprev;                  // [stage 0][3]: the start (after the init pass)
#define R2_STAMP(stage, k) if (R2DBG && (p.debug & 64) && p.ts) { const unsigned long long t_ = __builtin_readcyclecounter(); if (stamp) p.ts[(size_t)blockIdx.x * 32 + (stage) * 4 + (k)] = t_ - tprev; tprev = t_; }

    for (int s = 1; s < 8; ++s) {
        const int layer = s - 1;
        f32x16 acc[MI];
        float zero = 0.0f;
        asm volatile("" : "+v"(zero));                  // the accumulators start HERE (not hoisted over the previous stage's row pass)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = zero;

        constexpr int RSTEP = 512 / NCH;                            // rows per trip of the whole workgroup (32 / 64)
        constexpr int RB = 7;                                       // row chunks per thread per group
        constexpr int NPRE = (CF::TMAX / RSTEP + RB - 1) / RB * RB; // row chunks per thread: 14 / 7 (slices: 7 / 7)
        const bool noglob = R2DBG && (p.debug & 1);

        // every row chunk of c_{s+1} is requested before the last tap's MFMAs and consumed in the row pass below
        u32x4 cpre[NPRE];
        for (int tap = 0; tap < 3; ++tap) {
            if (tap == 2 && s < 7 && !noglob) {
                int tid_p = tid;
                asm volatile("" : "+v"(tid_p));
                const int row0 = tid_p / NCH, ch0 = tid_p % NCH;
                const char* h1s = reinterpret_cast<const char*>(H1) + (s + 1) * CW * 2;
#pragma unroll
                for (int k = 0; k < NPRE; ++k) {
                    const int rowc = min(row0 + k * RSTEP, T - 1);
                    cpre[k] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(h1s + ((uint32_t)rowc * (uint32_t)p.ld + (uint32_t)ch0 * 8u) * 2u));
                }
            }
            const bool more = !(s == 7 && tap == 2);
            if (more && !(R2DBG && (p.debug & 4))) { if (tap < 2) wload(layer, tap + 1); else wload(layer + 1, 0); }
            const int delta = (tap - 1) * p.dil;
            // (per-tap copies the optimiser cannot see through: hoisted out of the stage loop, the swizzled offsets of every
            //  (K step, tile) pair cost more registers than the double-buffered fragments and spill)
            int fh_t = fh, fr_t = fr;
            asm volatile("" : "+v"(fh_t), "+v"(fr_t));
            int rbase[MI], rsw[MI];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int t = (wq + CF::MW * i) * 32 + fr_t + delta;
                // reflect at the ends of the UTTERANCE; rows outside this slice's image are clamped (they only feed halo rows
                // that are already outside the dependency cone of the core)
                const int row = min(max(reflect_idx(l0 + t, Tg) - l0, 0), T - 1);
                rbase[i] = row * ROWB;
                rsw[i] = CF::swz(row);
            }
            const int wrow = (wn * 32 + fr_t);
            const int wbase = wrow * ROWB, wsw = CF::swz(wrow);
            // ONE straight-line body: all MI m-tiles (the compiler keeps the next K step's fragment reads in flight under the MFMAs).
            // Tiles that start at or past T multiply clamped rows and are dropped by the epilogue (with T = 401 that is one tile
            // of the wq = 1 waves, which wait at the barrier for the wq = 0 waves anyway; short utterances waste MFMAs on a kernel
            // that is latency-bound for them).  Measured alternatives: a per-tile `if (tile < T)` around each MFMA puts every MFMA
            // in its own basic block — fragment read, full lgkmcnt wait, one MFMA, 56 times per tap (440 TFLOP/s); a second body
            // for short utterances beside this one makes the register allocator spill in the row pass below.
            if (!(R2DBG && (p.debug & 2))) {
                constexpr int NK = CW / 16;
#pragma unroll
                for (int kk = 0; kk < NK; ++kk) {
                    const int ch = 2 * kk + fh_t;
                    const bf16x8 wf = *reinterpret_cast<const bf16x8*>(Wt + wbase + ((ch ^ wsw) << 4));
                    bf16x8 xf[MI];
#pragma unroll
                    for (int i = 0; i < MI; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(U + rbase[i] + ((ch ^ rsw[i]) << 4));
#pragma unroll
                    for (int i = 0; i < MI; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf[i], acc[i], 0, 0, 0);
                }
            }
            __syncthreads();                                    // Wt (and, after tap 2, U) are free
            if (tap < 2) { if (!(R2DBG && (p.debug & 4))) wstore(); __syncthreads(); }
        }

        R2_STAMP(s, 0)
        // ---- epilogue: acc[i][4g+e] = (t = (wq + MW*i)*32 + fr, n = wn*32 + 8g + 4fh + e) -> U ------
        int fh_e = fh, fr_e = fr;
        asm volatile("" : "+v"(fh_e), "+v"(fr_e));
        if (!(R2DBG && (p.debug & 8)))
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = wn * 32 + 8 * g + 4 * fh_e;
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias[layer] + n);
            const f32x4 sc4 = *reinterpret_cast<const f32x4*>(p.scale[layer] + n);
            const f32x4 sh4 = *reinterpret_cast<const f32x4*>(p.shift[layer] + n);
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int t = (wq + CF::MW * i) * 32 + fr_e;
                if (t < T) {
                    typedef bf16_t bf16x4 __attribute__((ext_vector_type(4)));
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = static_cast<bf16_t>(fmaf(fmaxf(acc[i][4 * g + e] + b4[e], 0.0f), sc4[e], sh4[e]));
                    const int c8 = n >> 2;                      // 8-byte chunk index inside the row
                    *reinterpret_cast<bf16x4*>(U + t * ROWB + ((((c8 >> 1) ^ CF::swz(t)) << 4) | ((c8 & 1) << 3))) = o;
                }
            }
        }
        if (s < 7 && !(R2DBG && (p.debug & 4))) wstore();      // next layer's tap-0 slab (loaded during tap 2)
        __syncthreads();
        R2_STAMP(s, 1)

        // ---- y_s -> H2 (whole rows), U <- y_s + c_{s+1} -------------------------------------------------
        // seven row chunks per thread at a time: their c_{s+1} loads are all in flight before the first is consumed (one
        // chunk per trip exposed a full HBM round trip per trip: ~0.8 us x 13 trips x 7 stages)
        if (R2DBG && (p.debug & 32)) {
            for (int c = tid; c < T * NCH; c += 512) {
                const int row = c / NCH, ch = c % NCH;
                char* up = U + row * ROWB + ((ch ^ CF::swz(row)) << 4);
                const bf16x8 y = *reinterpret_cast<const bf16x8*>(up);
                if (row >= s0 && row < s1) *reinterpret_cast<bf16x8*>(H2 + (int64_t)row * p.ld + s * CW + ch * 8) = y;
                if (s < 7) {
                    const bf16x8 cn = *reinterpret_cast<const bf16x8*>(H1 + (int64_t)row * p.ld + (s + 1) * CW + ch * 8);
                    bf16x8 u;
#pragma unroll
                    for (int e = 0; e < 8; ++e) u[e] = static_cast<bf16_t>(static_cast<float>(y[e]) + static_cast<float>(cn[e]));
                    *reinterpret_cast<bf16x8*>(up) = u;
                }
            }
        } else {
            int tid_r = tid;
            asm volatile("" : "+v"(tid_r));
            const int row0 = tid_r / NCH, ch0 = tid_r % NCH;
            const uint32_t goff0 = ((uint32_t)row0 * (uint32_t)p.ld + (uint32_t)ch0 * 8u) * 2u;
            const uint32_t gstep = (uint32_t)RSTEP * (uint32_t)p.ld * 2u;
            char* h2s = reinterpret_cast<char*>(H2) + s * CW * 2;
#pragma unroll
            for (int g = 0; g < NPRE / RB; ++g) {                   // every row chunk of c_{s+1} was requested before the last tap
                const int it0 = g * RB;
                if (it0 * RSTEP < T) {
#pragma unroll
                    for (int k = 0; k < RB; ++k) {
                        const int row = row0 + (it0 + k) * RSTEP;
                        if (row < T) {
                            char* up = U + row * ROWB + ((ch0 ^ CF::swz(row)) << 4);
                            const bf16x8 y = *reinterpret_cast<const bf16x8*>(up);
                            if (!noglob && row >= s0 && row < s1) *reinterpret_cast<bf16x8*>(h2s + goff0 + (uint32_t)(it0 + k) * gstep) = y;
                            if (s < 7 && !noglob) {
                                const bf16x8 cv = __builtin_bit_cast(bf16x8, cpre[it0 + k]);
                                bf16x8 u;
#pragma unroll
                                for (int e = 0; e < 8; ++e) u[e] = static_cast<bf16_t>(static_cast<float>(y[e]) + static_cast<float>(cv[e]));
                                *reinterpret_cast<bf16x8*>(up) = u;
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();
        R2_STAMP(s, 2)
    }
#undef R2_STAMP
}

template <int CW, int MIT>
hipError_t launch_cw(const Res2Params& p, int grid, hipStream_t stream) {
    static DeviceOnce attr;
    constexpr int lds = R2Cfg<CW, MIT>::LDS;
    if (hipError_t e = set_max_dynamic_lds(attr, reinterpret_cast<const void*>(res2net_chain_kernel<CW, MIT>), lds)) return e;
    hipLaunchKernelGGL((res2net_chain_kernel<CW, MIT>), dim3(grid), dim3(512), lds, stream, p);
    return hipGetLastError();
}

}  // namespace

// time slices for small batches: the LARGEST S in 2 .. 7 whose slices (core + two halos of 7 * dil frames) fit the short LDS image, whose
// core is longer than its halo and whose S * B workgroups are all resident at once (80 / 48 KiB of LDS each); 1 = whole utterances
int res2net_chain_slices(int B, int C, int T, int dil, int num_cu) {
    const int cw = C / 8;
    const int tmax_s = cw == 128 ? R2Cfg<128, 3>::TMAX : R2Cfg<64, 2>::TMAX;
    // the LARGEST slice count whose workgroups still fit the chip at once and whose core is longer than its halo: a slice's rows (core + 14 dil
    // of halo) set the launch's length (B = 20: 3 slices 178 us for the three launches, 5 - 7 slices 158 - 160; outputs bit-identical for any count)
    int best = 1;
    for (int S = 2; S <= 7; ++S) {
        const int tc = (T + S - 1) / S;
        if ((int64_t)S * B > (int64_t)num_cu) break;
        if (tc + 14 * dil <= tmax_s && tc > 14 * dil) best = S;
    }
    return best;
}

bool res2net_chain_supported(int C, int T, int dil, int Kp) {
    const int cw = C / 8;
    return (cw == 64 || cw == 128) && T <= R2_TMAX && T > 2 * dil && Kp == 3 * cw;
}

hipError_t launch_res2net_chain(const Res2Params& p_in, int B, int C, hipStream_t stream) {
    if (!res2net_chain_supported(C, p_in.T, p_in.dil, p_in.Kp) || B <= 0 || p_in.ld % 8 != 0) return hipErrorInvalidValue;
    Res2Params p = p_in;
    if (p.slices > 1) {
        const int tmax_s = C / 8 == 128 ? R2Cfg<128, 3>::TMAX : R2Cfg<64, 2>::TMAX;
        // a slice count that does not fit (the developer option r2_slices forces one): the nearest smaller count that does, down to whole
        // utterances — never a failed forward
        while (p.slices > 1) {
            p.Tc = (p.T + p.slices - 1) / p.slices;
            if (p.Tc + 14 * p.dil <= tmax_s && p.Tc > 14 * p.dil) break;
            --p.slices;
        }
        if (p.slices > 1) return C / 8 == 128 ? launch_cw<128, 3>(p, B * p.slices, stream) : launch_cw<64, 2>(p, B * p.slices, stream);
    }
    p.slices = 1;
    return C / 8 == 128 ? launch_cw<128, 7>(p, B, stream) : launch_cw<64, 4>(p, B, stream);
}

}  // namespace svhip
