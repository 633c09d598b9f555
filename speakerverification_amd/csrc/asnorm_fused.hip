// asnorm_fused.hip — AS-norm cohort statistics WITHOUT the N x K score matrix (gfx950).
//
// Reference: src/utils.py:142-146 — per embedding e: S = cohort @ e, sort descending, keep [:top], mean / population std.
// The slab path (score.hip + the fp32 GEMM) wrote all N x K scores to HBM and read them back (28.8 GB each way for 1.2 M
// embeddings against 5 994 cohort speakers).  Here the scores live only in MFMA accumulators:
//
//   * a wave owns 32 embeddings as the B operand of v_mfma_f32_32x32x2_f32 (exact fp32), D / 2 VGPRs per lane, loaded once;
//     cohort rows stream through LDS in blocks of 32 (the A operand; one 16-byte-per-lane LDS-DMA image per block, XOR-swizzled
//     on the source chunk so that the ds_read_b128 lane groups are conflict-free), shared by the workgroup's four waves;
//   * with the cohort as A, a lane's 16 accumulators are 16 cohort scores of ONE embedding (C/D map: column = lane & 31), so the
//     selection is lane-local: a score above the embedding's threshold tau is appended to that lane's candidate list in a
//     handle-owned buffer (two lists of 256 per embedding: lanes l and l + 32), everything else is dropped;
//   * tau needs the row's first two score moments BEFORE the pass.  They are exact and cheap:  mean_k(e . c_k) = e . cbar and
//     mean_k (e . c_k)^2 = e^T M e with M = C^T C / K (D x D), so M's rows and cbar go through the same MFMA loop as D / 32 + 1
//     leading pseudo-cohort blocks; tau = mean + z sd with z the normal quantile that leaves ~1.6 top candidates;
//   * asnorm_cand_stats_kernel then selects the exact top-`top` of each embedding's <= 512 candidates (score_select.h) —
//     every score above tau is a candidate, so the answer is exact whenever top <= count and no list overflowed; the rare
//     embedding for which that fails (cohort scores far from normal) is flagged and redone by the slab path.
//
// Work per embedding: 2 K D FLOP on the fp32 matrix pipe (157 TFLOP/s peak), ~K/32 x 24 KB of LDS-DMA per 128 embeddings from an
// L2 / Infinity-Cache resident cohort, ~1.6 top x 4 bytes of candidate stores.
#include <type_traits>

#include "common.h"
#include "kernels.h"
#include "score_select.h"

namespace svhip {

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

constexpr int AF_CAPL = ASNORM_CAND_PER_LANE;      // candidate slots per lane (2 lanes per embedding)

template <int D>
__global__ __launch_bounds__(256, 2) void asnorm_fused_kernel(AsnormFusedParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int CH = D / 4;                       // 16-byte chunks per cohort row
    constexpr int BLK = 32 * D * 4;                 // one block of 32 cohort rows
    constexpr int NDMA = 32 * CH / 256;             // DMA instructions per thread per block
    constexpr int NP = D / 32 + 1;                  // pseudo-cohort blocks: rows of M, then cbar
    static_assert(D % 64 == 0 && (32 * CH) % 256 == 0, "block image must be whole 4 KiB DMA rounds with 16-aligned chunk groups");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;         // embedding within the wave's 32 / half of the k range (and of the cohort rows of a block)
    const int64_t row = (int64_t)blockIdx.x * 128 + wave * 32 + j;
    const bool valid = row < p.N;
    const float* __restrict__ erow = p.E + (valid ? row : p.N - 1) * D;

    // B operand: e[k], k = h * D/2 + s, for MFMA k-step s (both operands use the same k assignment, so any is as good as 0, 1, 2 ..)
    float eb[D / 2];
#pragma unroll
    for (int t = 0; t < D / 8; ++t) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(erow + h * (D / 2) + t * 4);
#pragma unroll
        for (int u = 0; u < 4; ++u) eb[t * 4 + u] = v[u];
    }

    const int nc = (p.K + 31) / 32;
    const int nb = NP + nc;
    // per-lane source offsets of the NDMA chunks this lane moves per block (block-invariant: row within the block, swizzled chunk)
    int srow[NDMA], scol[NDMA];
#pragma unroll
    for (int q = 0; q < NDMA; ++q) {
        const int pidx = q * 256 + tid;               // position of this lane's chunk in the block image
        const int i = pidx / CH, cs = pidx - i * CH;
        srow[q] = i;
        scol[q] = (cs ^ (i & 15)) * 4;                // the image is lane-linear: the swizzle goes on the source
    }
    auto issue = [&](int b, int buf) {
        const bool pseudo = b < NP;
        const float* base = pseudo ? p.MB : p.cohort;
        const int r0 = pseudo ? b * 32 : (b - NP) * 32;
        const int limit = pseudo ? D + 31 : p.K - 1;
        const float* bb = base + (int64_t)r0 * D;
        const bool ragged = r0 + 31 > limit;          // wave-uniform: only the cohort's last block clamps its rows
#pragma unroll
        for (int q = 0; q < NDMA; ++q) {
            const int i = ragged ? min(srow[q], limit - r0) : srow[q];
            const float* s = bb + i * D + scol[q];
            __builtin_amdgcn_global_load_lds((gbl_void*)s, (lds_void*)(smem + buf * BLK + (q * 256 + wave * 64) * 16), 16, 0, 0);
        }
    };
    // A fragments (16-byte reads: four k-steps each) are requested NG1 groups ahead of their MFMAs (left alone hipcc issues each
    // ds_read right in front of the wait of the four MFMAs that use it).
    // Read-ahead depth, measured on one box with two builds (tools/build_variant.sh -DAF_NG1=n; 1.2 M x 5 994, ms end to end):
    // none 26.1 - 26.4 (another box), 1 group 26.7 - 26.9, half a block (12) 27.5 - 29.3: the partner wave already covers the LDS
    // latency, and the deeper variants only cost registers.  PMC of this kernel (profiles/r03_asnorm_pmc.txt): matrix pipe busy
    // 0.80 of the CU cycles at 2.13 - 2.23 GHz, no LDS bank conflicts, LDS array 5 % active.
#ifdef AF_NG1
    constexpr int NG = CH / 2, NG1 = AF_NG1;
#else
    constexpr int NG = CH / 2, NG1 = 1;
#endif
    f32x4 af[NG];
    auto rd = [&](int buf, int t) {
        return *reinterpret_cast<const f32x4*>(smem + buf * BLK + j * (D * 4) + (((h * NG + t) ^ (j & 15)) << 4));      // cohort row j of the block (A: row = lane & 31)
    };
    auto read_first = [&](int buf) {
#pragma unroll
        for (int t = 0; t < NG1; ++t) af[t] = rd(buf, t);
    };
    auto mfma_block = [&](int buf) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int t = 0; t < NG; ++t) {
            if (t + NG1 < NG) af[t + NG1] = rd(buf, t + NG1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[t][u], eb[t * 4 + u], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
        return acc;
    };

    float second = 0.0f, tau = 0.0f;
    int cnt = 0;
    float* candl = p.cand + ((valid ? row : 0) * 2 + h) * AF_CAPL;
    // acc[r] = score of cohort row 32 b' + (r & 3) + 8 (r >> 2) + 4 h against embedding j
    auto process = [&](const f32x16& a, int b) {
        if (b < NP - 1) {               // rows of M: (M e)_i . e_i
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 ev = *reinterpret_cast<const f32x4*>(erow + 32 * b + 8 * g + 4 * h);
#pragma unroll
                for (int u = 0; u < 4; ++u) second = fmaf(a[4 * g + u], ev[u], second);
            }
        } else if (b == NP - 1) {       // row 0 of the last pseudo block is cbar: the mean lives in register 0 of the h = 0 lanes
            const float m = __shfl(a[0], j, 64);
            const float sec = second + __shfl_xor(second, 32, 64);
            tau = m + (p.zrow ? p.zrow[valid ? row : 0] : p.z) * sqrtf(fmaxf(sec - m * m, 0.0f));
        } else {
            const int kb = (b - NP) * 32 + 4 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = kb + (r & 3) + 8 * (r >> 2);
                const float v = a[r];
                if (i < p.K && v > tau) {
                    if (cnt < AF_CAPL && valid) candl[cnt] = v;
                    ++cnt;
                }
            }
        }
    };

#ifdef AF_STAGGER
    if ((blockIdx.x >> 8) & 1) __builtin_amdgcn_s_sleep(AF_STAGGER);      // developer A/B: co-resident workgroups half a block apart (measured: +-0)
#endif
    issue(0, 0);
    f32x16 accp;
#pragma unroll
    for (int r = 0; r < 16; ++r) accp[r] = 0.0f;
    for (int b = 0; b < nb; ++b) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // block b has landed (and the candidate stores of one block ago, long since)
        __syncthreads();                                       // ... for every wave; nobody still reads the other buffer
        if (b + 1 < nb) issue(b + 1, (b + 1) & 1);
        read_first(b & 1);
        __builtin_amdgcn_sched_barrier(0);
        if (b > 0) process(accp, b - 1);                       // selection of the previous block (the first reads land meanwhile)
        accp = mfma_block(b & 1);
    }
    process(accp, nb - 1);
    if (valid) p.cnt[row * 2 + h] = cnt;
}

// ---- the same kernel on SIX bf16 MFMAs per product block ("x6"): fp32-grade scores at 2.7 x the fp32 matrix rate ----------------
// An fp32 value splits EXACTLY into three bf16 parts, v = h + m + l (8 + 8 + 8 mantissa bits), so a product is the nine partial
// products of the parts; the three smallest (m l, l m, l l: <= 2^-26 of |a b|) are dropped and the other six are bf16 MFMAs with
// exact fp32 products and fp32 accumulation: h h, h m, m h, h l, l h, m m.  A score (|e| = |c| = 1) is then within 1.5e-8 of the
// exact-fp32-MFMA kernel's plus the usual accumulation rounding — three orders below anything AS-norm can see (the 2^-17 of the
// three-product form is NOT: 2e-4 on the normalised scores) — while v_mfma_f32_32x32x16_bf16 does 16 k values in the 32
// cycles v_mfma_f32_32x32x2_f32 needs for one: six of them per 16 k against eight fp32 MFMAs of 64 cycles, 192 against 512 cycles.
// Same structure as above (embeddings = B operand in registers, cohort blocks through LDS, lane-local selection, exact moments as
// leading pseudo-cohort blocks); the cohort image (pseudo rows first, then the K cohort rows) is split into three bf16 planes once
// per call (split3_planes_kernel).  D = 192 only: the B operand is 144 VGPRs.
//
// NPL = 2 ("h3", round 4, the default): TWO planes of IEEE half and THREE fp16 MFMAs per product block (hi.hi + hi.lo + lo.hi).  A half
// plane carries 11 significant bits, so hi + lo holds 22 (2^-23 relative) while lo is a normal half, and below |v| = 2^-3 — where
// the components of unit embeddings live — lo is a subnormal half with an ABSOLUTE quantum of 2^-24: a component is represented to
// 3e-8 absolute, a score of unit vectors to ~4e-8 (measured against the float64 oracle: tests), the same class as x6, at half the MFMAs,
// two thirds of the LDS bytes and 96 instead of 144 operand VGPRs (D = 256 fits as well).
template <int D, int NPL>
__global__ __launch_bounds__(256, 2) void asnorm_fused6_kernel(AsnormFusedParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int CH = D / 8;                       // 16-byte chunks (8 values) per row of a plane
    constexpr int PL = 32 * D * 2;                  // one plane of a block of 32 rows
    constexpr int BLK = NPL * PL;
    constexpr int NDMA = NPL * 32 * CH / 256;       // DMA instructions per thread per block
    constexpr int NP = D / 32 + 1;                  // pseudo-cohort blocks: rows of M, then cbar
    constexpr int NS = D / 16;                      // MFMA k steps per block
    static_assert(D % 64 == 0 && (NPL * 32 * CH) % 256 == 0 && CH % 8 == 0, "block image: whole DMA rounds, chunk groups of 8");
    static_assert(NPL == 2 || NPL == 3, "two half planes or three bf16 planes");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int64_t row = (int64_t)blockIdx.x * 128 + wave * 32 + j;
    const bool valid = row < p.N;
    // (per-lane addresses are kept as 32-bit offsets from the scalar bases: every VGPR counts here)
    const uint32_t eoff = (uint32_t)((valid ? row : p.N - 1) * D);                  // < 2^31 elements per launch (host check)
    const float* __restrict__ erow = p.E + eoff;

    // B operand: embedding j, k = 16 s + 8 h .. + 7 for k step s, in three bf16 parts (or two half parts: bh, bm)
    bf16x8 bh[NS], bm[NS], bl[NPL == 3 ? NS : 1];
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(erow + 16 * s_ + 8 * h);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(erow + 16 * s_ + 8 * h + 4);
        if (NPL == 3) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float v = u < 4 ? v0[u] : v1[u - 4];
                const bf16_t a = static_cast<bf16_t>(v);
                const float r1 = v - static_cast<float>(a);
                const bf16_t b = static_cast<bf16_t>(r1);
                bh[s_][u] = a;
                bm[s_][u] = b;
                bl[s_][u] = static_cast<bf16_t>(r1 - static_cast<float>(b));
            }
        } else {
            f16x8 a8, b8;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float v = u < 4 ? v0[u] : v1[u - 4];
                const f16_t a = static_cast<f16_t>(v);
                a8[u] = a;
                b8[u] = static_cast<f16_t>(v - static_cast<float>(a));
            }
            bh[s_] = __builtin_bit_cast(bf16x8, a8);
            bm[s_] = __builtin_bit_cast(bf16x8, b8);
        }
    }

    const int rows_total = NP * 32 + p.K;
    const int nb = (rows_total + 31) / 32;
    const char* planes = reinterpret_cast<const char*>(p.planes);
    const int64_t plane_bytes = (int64_t)rows_total * D * 2;
    // the image is lane-linear (16 bytes per lane): position -> (plane, row, chunk slot); the swizzle goes on the source chunk
    auto issue = [&](int b, int buf) {
        const int r0 = b * 32;
        const int limit = rows_total - 1 - r0;          // the last block clamps its rows
#pragma unroll
        for (int q = 0; q < NDMA; ++q) {
            const int pidx = q * 256 + tid;
            const int pl = pidx / (32 * CH), rem = pidx - pl * (32 * CH);
            const int i = rem / CH, cs = rem - i * CH;
            const int c = (cs & ~7) | ((cs ^ (i >> 1)) & 7);
            const char* src = planes + pl * plane_bytes + ((int64_t)(r0 + min(i, limit)) * D + c * 8) * 2;
            __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(smem + buf * BLK + (q * 256 + wave * 64) * 16), 16, 0, 0);
        }
    };
    auto rd = [&](int buf, int pl, int s_) {          // A fragment: cohort row j of the block, k = 16 s + 8 h .. + 7
        const int cs = 2 * s_ + h;
        return *reinterpret_cast<const bf16x8*>(smem + buf * BLK + pl * PL + j * (D * 2) + (((cs & ~7) | ((cs ^ (j >> 1)) & 7)) << 4));
    };
    auto mfma_block = [&](int buf) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        // A fragments are read ONE k step ahead of their six MFMAs, fenced (left alone hipcc issues each ds_read right in front of
        // the wait of the MFMAs that use it, and a wave then sits out the LDS latency every k step whenever its SIMD partner is
        // not in its own MFMA phase: matrix pipe busy 0.58)
        bf16x8 ah = rd(buf, 0, 0), am = rd(buf, 1, 0), al = NPL == 3 ? rd(buf, 2, 0) : ah;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            bf16x8 nh = ah, nm = am, nl = al;
            if (s_ + 1 < NS) { nh = rd(buf, 0, s_ + 1); nm = rd(buf, 1, s_ + 1); if (NPL == 3) nl = rd(buf, 2, s_ + 1); }
            __builtin_amdgcn_sched_barrier(0);
            if (NPL == 3) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[s_], acc, 0, 0, 0);      // small terms first
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[s_], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm[s_], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh[s_], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm[s_], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[s_], acc, 0, 0, 0);
            } else {            // (planes: h = hi, m = lo)
                acc = Half16<f16_t>::mfma32(am, bh[s_], acc);
                acc = Half16<f16_t>::mfma32(ah, bm[s_], acc);
                acc = Half16<f16_t>::mfma32(ah, bh[s_], acc);
            }
            __builtin_amdgcn_sched_barrier(0);
            ah = nh; am = nm; al = nl;
        }
        __builtin_amdgcn_s_setprio(0);
        return acc;
    };

    float second = 0.0f, tau = 0.0f;
    int cnt = 0;
    const uint32_t coff = (uint32_t)(((valid ? row : 0) * 2 + h) * AF_CAPL);
    // acc[r] = score of image row 32 b + (r & 3) + 8 (r >> 2) + 4 h against embedding j
    auto process = [&](const f32x16& a, int b) {
        if (b < NP - 1) {               // rows of M: (M e)_i . e_i
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 ev = *reinterpret_cast<const f32x4*>(p.E + eoff + 32 * b + 8 * g + 4 * h);
#pragma unroll
                for (int u = 0; u < 4; ++u) second = fmaf(a[4 * g + u], ev[u], second);
            }
        } else if (b == NP - 1) {       // row 0 of the last pseudo block is cbar: the mean lives in register 0 of the h = 0 lanes
            const float m = __shfl(a[0], j, 64);
            const float sec = second + __shfl_xor(second, 32, 64);
            tau = m + (p.zrow ? p.zrow[valid ? row : 0] : p.z) * sqrtf(fmaxf(sec - m * m, 0.0f));
        } else {
            const int kb = (b - NP) * 32 + 4 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = kb + (r & 3) + 8 * (r >> 2);
                const float v = a[r];
                if (i < p.K && v > tau) {
                    if (cnt < AF_CAPL && valid) p.cand[coff + cnt] = v;
                    ++cnt;
                }
            }
        }
    };

    issue(0, 0);
    f32x16 accp;
#pragma unroll
    for (int r = 0; r < 16; ++r) accp[r] = 0.0f;
    for (int b = 0; b < nb; ++b) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // block b has landed (and the candidate stores of one block ago, long since)
        __syncthreads();                                       // ... for every wave; nobody still reads the other buffer
        if (b + 1 < nb) issue(b + 1, (b + 1) & 1);
        // the selection of the PREVIOUS block goes first: its candidate stores are then a whole MFMA phase old at the next wait
        // (issued after the MFMAs they sat right in front of that wait: 18.8 ms for 1.2 M embeddings)
        if (b > 0) process(accp, b - 1);
        __builtin_amdgcn_sched_barrier(0);
        accp = mfma_block(b & 1);
    }
    process(accp, nb - 1);
    if (valid) p.cnt[row * 2 + h] = cnt;
}

// ---- operand scaling of the half-plane forms (round 5) --------------------------------------------------------------------------
// IEEE-half planes have fp16's exponent range: a component beyond 65504 overflows, one below 2^-14 loses bits of its hi part, and lo is a
// subnormal (2^-24 absolute) below 2^-3.  Embeddings are not always unit vectors (the reference scores raw embeddings when `normalize` is
// off: src/model.py:421, utils.py:142-150), so every operand is brought to a fixed magnitude by an EXACT power of two before it is split:
// a row of the register operand by its own max |x| (found in the kernel, per row), the streamed operand by the max |x| of the whole matrix
// (absmax_bits_kernel -> a device word the split pass and the kernels read: no host round trip).  max |x| lands in [64, 128): hi is exact
// to 11 bits and lo a normal half for every component down to 2^-10 of the largest; squares (the AS-norm moment rows) stay below 2^14.
// Products and sums are formed on the scaled values — binary floating point is scale-invariant, so the fp32 accumulation rounds exactly
// as it would have — and the result is multiplied back by the exact inverse.  Zero rows keep scale 1; inf / NaN stay inf / NaN.
// (pow2_scale_of_bits, pow2_inverse, finite_abs_bits: common.h — round 6: the F32X3 network input uses them too)
// max |x| over the FINITE elements of a matrix as the bit pattern of a non-negative float (ordered like an unsigned integer): ABSMAX_WGS
// workgroups leave one partial each in part[0 .. ABSMAX_WGS) (no atomics, no zero-fill launch); the consumer's first kernel folds them
// (absmax_fold: 256 loads per workgroup) and its workgroup 0 publishes the result in part[-1] = *pscale for the kernels after it
constexpr int ABSMAX_WGS = 256;
__global__ __launch_bounds__(256) void absmax_bits_kernel(const float* __restrict__ X, int64_t n, uint32_t* __restrict__ part) {
    __shared__ uint32_t wm[4];
    uint32_t m = 0;
    const int64_t n4 = n >> 2;
    const u32x4* __restrict__ X4 = reinterpret_cast<const u32x4*>(X);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const u32x4 v = X4[i];
        m = max(max(m, finite_abs_bits(v[0])), max(max(finite_abs_bits(v[1]), finite_abs_bits(v[2])), finite_abs_bits(v[3])));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = max(m, finite_abs_bits(__float_as_uint(X[(n4 << 2) + threadIdx.x])));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = max(max(wm[0], wm[1]), max(wm[2], wm[3]));
}
// (called by every thread of a 256-thread workgroup)
__device__ __forceinline__ uint32_t absmax_fold(const uint32_t* __restrict__ part) {
    __shared__ uint32_t fm[4];
    uint32_t m = part[threadIdx.x];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0) fm[threadIdx.x >> 6] = m;
    __syncthreads();
    return max(max(fm[0], fm[1]), max(fm[2], fm[3]));
}

// ---- the two-half-plane form on v_mfma_f32_16x16x32_f16 ("h3w", round 4, the default) -------------------------------------------------
// Same arithmetic as asnorm_fused6_kernel<D, 2> (hi.hi + hi.lo + lo.hi on IEEE-half planes, fp32 accumulate), on the 16-wide MFMA: on random
// data the chip holds a higher clock under the 16x16x32 instruction than under 32x32x16 (bare loops: 1.87 against 1.63 PFLOP/s,
// tools/mfma_rate.hip; MI355X_MICROARCH.md, DVFS give-back), and this kernel runs at the pace of its MFMAs.  A wave still owns 32 embeddings
// and walks the cohort in blocks of 32 rows: 2 row groups x 2 embedding groups of 16 per k step of 32.  Accumulator layout: lane (c = lane & 15,
// q = lane >> 4) holds the scores of cohort rows 16 rg + 4 q + e (e = 0..3) against embedding 16 eg + c — so a lane selects for TWO embeddings,
// and an embedding's candidates come from FOUR lanes: four candidate lists of ASNORM_CAND_PER_LANE / 2 per embedding (AsnormFusedParams::nlists).
template <int D>
__global__ __launch_bounds__(256, 2) void asnorm_h3w_kernel(AsnormFusedParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int CH = D / 8;                       // 16-byte chunks (8 values) per row of a plane
    constexpr int PL = 32 * D * 2;                  // one plane of a block of 32 rows
    constexpr int BLK = 2 * PL;
    constexpr int NDMA = 2 * 32 * CH / 256;         // DMA instructions per thread per block
    constexpr int NP = D / 32 + 1;                  // pseudo-cohort blocks: rows of M, then cbar
    constexpr int NS = D / 32;                      // MFMA k steps (32 wide) per block
    constexpr int CAPQ = AF_CAPL / 2;               // candidate slots per (embedding, q)
    static_assert(D % 64 == 0 && (2 * 32 * CH) % 256 == 0 && CH % 8 == 0, "block image: whole DMA rounds, chunk groups of 8");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, q = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * 128 + wave * 32 + c;            // embedding of group 0 (group 1: + 16)
    bool valid[2];
    uint32_t eoff[2];                                                          // < 2^31 elements per launch (host check)
#pragma unroll
    for (int eg = 0; eg < 2; ++eg) {
        const int64_t r = row0 + 16 * eg;
        valid[eg] = r < p.N;
        eoff[eg] = (uint32_t)((valid[eg] ? r : p.N - 1) * D);
    }

    // B operand: embedding (16 eg + c), k = 32 s + 8 q .. + 7, as half hi | lo parts of the row SCALED by a power of two (its max |x| in
    // [64, 128): "operand scaling" above).  With p.pscale the planes are scaled too (cohort and mean row by sC, moment rows by sC^2): every
    // score of this kernel lives in the scaled domain — the threshold is formed and compared there, the candidates are stored there — and
    // the candidate kernel multiplies a row's mean and deviation back by p.rowscale[row] (exact).
    bf16x8 bh[2][NS], bl[2][NS];
    float se[2] = {1.0f, 1.0f}, unscale[2] = {1.0f, 1.0f};
#pragma unroll
    for (int eg = 0; eg < 2; ++eg) {
        if (p.pscale) {
            uint32_t m = 0;
#pragma unroll
            for (int s_ = 0; s_ < NS; ++s_) {
                const u32x4 w0 = *reinterpret_cast<const u32x4*>(p.E + eoff[eg] + 32 * s_ + 8 * q);
                const u32x4 w1 = *reinterpret_cast<const u32x4*>(p.E + eoff[eg] + 32 * s_ + 8 * q + 4);
#pragma unroll
                for (int u = 0; u < 4; ++u) m = max(m, max(finite_abs_bits(w0[u]), finite_abs_bits(w1[u])));
            }
            m = max(m, (uint32_t)__shfl_xor((int)m, 16, 64));
            m = max(m, (uint32_t)__shfl_xor((int)m, 32, 64));
            se[eg] = pow2_scale_of_bits(m);
            unscale[eg] = pow2_inverse(se[eg]) * pow2_inverse(pow2_scale_of_bits(*p.pscale));
        }
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(p.E + eoff[eg] + 32 * s_ + 8 * q);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(p.E + eoff[eg] + 32 * s_ + 8 * q + 4);
            f16x8 a8, b8;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float v = (u < 4 ? v0[u] : v1[u - 4]) * se[eg];
                const f16_t a = static_cast<f16_t>(v);
                a8[u] = a;
                b8[u] = static_cast<f16_t>(v - static_cast<float>(a));
            }
            bh[eg][s_] = __builtin_bit_cast(bf16x8, a8);
            bl[eg][s_] = __builtin_bit_cast(bf16x8, b8);
        }
    }

    const int rows_total = NP * 32 + p.K;
    const int nb = (rows_total + 31) / 32;
    const char* planes = reinterpret_cast<const char*>(p.planes);
    const int64_t plane_bytes = (int64_t)rows_total * D * 2;
    // the block image is the one of asnorm_fused6_kernel: lane-linear DMA, the swizzle on the source chunk
    auto issue = [&](int b, int buf) {
        const int r0 = b * 32;
        const int limit = rows_total - 1 - r0;          // the last block clamps its rows
#pragma unroll
        for (int qq = 0; qq < NDMA; ++qq) {
            const int pidx = qq * 256 + tid;
            const int pl = pidx / (32 * CH), rem = pidx - pl * (32 * CH);
            const int i = rem / CH, cs = rem - i * CH;
            const int cc = (cs & ~7) | ((cs ^ (i >> 1)) & 7);
            const char* src = planes + pl * plane_bytes + ((int64_t)(r0 + min(i, limit)) * D + cc * 8) * 2;
            __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(smem + buf * BLK + (qq * 256 + wave * 64) * 16), 16, 0, 0);
        }
    };
    // A fragment: cohort row 16 rg + c of the block, k = 32 s + 8 q .. + 7 (chunk 4 s + q of the row).  The 16 lanes of one q read 16 rows at
    // one logical chunk: 8 keys x even / odd row (row stride 384 B = 1.5 bank periods at D = 192): 16 distinct 16-byte slots, no conflict
    auto rd = [&](int buf, int pl, int rg, int s_) {
        const int i = 16 * rg + c, cs = 4 * s_ + q;
        return *reinterpret_cast<const bf16x8*>(smem + buf * BLK + pl * PL + i * (D * 2) + (((cs & ~7) | ((cs ^ (i >> 1)) & 7)) << 4));
    };
    typedef f32x4 acc_t[2][2];
    auto mfma_block = [&](int buf, acc_t& acc) {
#pragma unroll
        for (int rg = 0; rg < 2; ++rg)
#pragma unroll
            for (int eg = 0; eg < 2; ++eg) acc[rg][eg] = f32x4{0.f, 0.f, 0.f, 0.f};
        // A fragments one k step ahead of their MFMAs, fenced (see asnorm_fused6_kernel)
        bf16x8 ah[2] = {rd(buf, 0, 0, 0), rd(buf, 0, 1, 0)}, al[2] = {rd(buf, 1, 0, 0), rd(buf, 1, 1, 0)};
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            bf16x8 nh[2] = {ah[0], ah[1]}, nl[2] = {al[0], al[1]};
            if (s_ + 1 < NS) {
#pragma unroll
                for (int rg = 0; rg < 2; ++rg) { nh[rg] = rd(buf, 0, rg, s_ + 1); nl[rg] = rd(buf, 1, rg, s_ + 1); }
            }
            __builtin_amdgcn_sched_barrier(0);
            // small terms first; the four accumulators of a term are independent
#pragma unroll
            for (int rg = 0; rg < 2; ++rg)
#pragma unroll
                for (int eg = 0; eg < 2; ++eg) acc[rg][eg] = Half16<f16_t>::mfma16(al[rg], bh[eg][s_], acc[rg][eg]);
#pragma unroll
            for (int rg = 0; rg < 2; ++rg)
#pragma unroll
                for (int eg = 0; eg < 2; ++eg) acc[rg][eg] = Half16<f16_t>::mfma16(ah[rg], bl[eg][s_], acc[rg][eg]);
#pragma unroll
            for (int rg = 0; rg < 2; ++rg)
#pragma unroll
                for (int eg = 0; eg < 2; ++eg) acc[rg][eg] = Half16<f16_t>::mfma16(ah[rg], bh[eg][s_], acc[rg][eg]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rg = 0; rg < 2; ++rg) { ah[rg] = nh[rg]; al[rg] = nl[rg]; }
        }
        __builtin_amdgcn_s_setprio(0);
    };

    float second[2] = {0.0f, 0.0f}, tau[2] = {0.0f, 0.0f};
    // A lane's candidate list.  The four lists of an embedding (one per q) are INTERLEAVED in its 4 * CAPQ-float region: element k of list q at
    // float k * 4 + q — the four lanes fill neighbouring dwords at about the same pace (16-byte groups complete within a block or two instead of
    // a lane's own eight-element sector over ten), and the statistics kernel finds the region dense up to 4 * (shortest list) and can stop its
    // search registers there.  pos = BYTE offset of the next free slot (16 bytes per element), lim = byte offset of the last slot.
    uint32_t coff[2], pos[2], lim[2];
#pragma unroll
    for (int eg = 0; eg < 2; ++eg) {
        coff[eg] = (uint32_t)((valid[eg] ? row0 + 16 * eg : 0) * 4 * CAPQ + q);
        pos[eg] = coff[eg] << 2;
        lim[eg] = (coff[eg] + (uint32_t)(CAPQ - 1) * 4u) << 2;
    }
    auto process = [&](const acc_t& a, int b) {
        if (b < NP - 1) {               // rows of M: (M e)_i . e_i, i = 32 b + 16 rg + 4 q + e
#pragma unroll
            for (int eg = 0; eg < 2; ++eg)
#pragma unroll
                for (int rg = 0; rg < 2; ++rg) {
                    const f32x4 ev = *reinterpret_cast<const f32x4*>(p.E + eoff[eg] + 32 * b + 16 * rg + 4 * q);
#pragma unroll
                    for (int u = 0; u < 4; ++u) second[eg] = fmaf(a[rg][eg][u], ev[u] * se[eg], second[eg]);      // (scaled domain: sC^2 sE^2 e^T M e)
                }
        } else if (b == NP - 1) {       // row 0 of the last pseudo block is cbar: the mean lives in register 0 of row group 0 of the q = 0 lanes
#pragma unroll
            for (int eg = 0; eg < 2; ++eg) {
                const float m = __shfl(a[0][eg][0], c, 64);
                float sec = second[eg];
                sec += __shfl_xor(sec, 16, 64);
                sec += __shfl_xor(sec, 32, 64);
                // (rows past N never hit: no separate validity test in the selection below)
                const float zz = p.zrow ? p.zrow[valid[eg] ? row0 + 16 * eg : 0] : p.z;
                tau[eg] = valid[eg] ? m + zz * sqrtf(fmaxf(sec - m * m, 0.0f)) : __builtin_inff();
            }
        } else {
            // The selection, 32 scores per lane per block.  It issues for the whole wave whenever ONE lane hits (4.7 % of the scores pass:
            // 95 % of the instructions see a hit somewhere), so what counts is its instruction count, not how rarely a lane stores.  Round 4's
            // form — row-bound test, threshold test, list-capacity test as nested ifs, a 64-bit store address — was ~7 vector + ~8 scalar
            // instructions and two branches per score: ~1.7 k issue cycles per block beside 1.15 k of MFMAs.  Now: one compare, a clamped slot
            // (an overflowing list is flagged by its count and never read, so its last slot may be overwritten), a 32-bit offset from the
            // scalar base, one masked store, one add; the row-bound test exists only in the cohort's last block.
            const int kb0 = (b - NP) * 32;
            char* cbase = reinterpret_cast<char*>(p.cand);
            auto select = [&](auto bounded) {                   // bounded: the cohort's last block, whose rows past K repeat row K - 1
#pragma unroll
                for (int eg = 0; eg < 2; ++eg)
#pragma unroll
                    for (int rg = 0; rg < 2; ++rg)
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const float v = a[rg][eg][u];
                            bool hit = v > tau[eg];
                            if (decltype(bounded)::value) hit = hit && (kb0 + 4 * q + 16 * rg + u < p.K);
                            // branch-free: the slot address is formed for every lane, the store runs under the hit mask as EXEC (written
                            // as `if (hit)` the compiler adds an s_cbranch_execz per score and lays the two paths out alternately:
                            // a taken branch on most of the 32 scores of a block)
                            // (the list position is kept as a BYTE offset: clamp, store, advance by one element on a hit — four vector instructions per score)
                            const uint32_t off = min(pos[eg], lim[eg]);
                            const unsigned long long mask = __ballot(hit);
                            unsigned long long saved;
                            asm volatile("s_and_saveexec_b64 %0, %1\n\tglobal_store_dword %2, %3, %4\n\ts_mov_b64 exec, %0"
                                         : "=&s"(saved) : "s"(mask), "v"(off), "v"(v), "s"(cbase) : "memory", "scc");
                            pos[eg] += hit ? 16u : 0u;
                        }
            };
            if (kb0 + 32 <= p.K) select(std::false_type{});     // (wave-uniform)
            else select(std::true_type{});
        }
    };

    issue(0, 0);
    acc_t accp;
#pragma unroll
    for (int rg = 0; rg < 2; ++rg)
#pragma unroll
        for (int eg = 0; eg < 2; ++eg) accp[rg][eg] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int b = 0; b < nb; ++b) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // block b has landed (and the candidate stores of one block ago, long since)
        __syncthreads();                                       // ... for every wave; nobody still reads the other buffer
        if (b + 1 < nb) issue(b + 1, (b + 1) & 1);
        if (b > 0) process(accp, b - 1);                       // the selection of the PREVIOUS block first (asnorm_fused6_kernel)
        __builtin_amdgcn_sched_barrier(0);
        mfma_block(b & 1, accp);
    }
    process(accp, nb - 1);
#pragma unroll
    for (int eg = 0; eg < 2; ++eg)
        if (valid[eg]) {
            p.cnt[(row0 + 16 * eg) * 4 + q] = (int32_t)((pos[eg] - (coff[eg] << 2)) >> 4);      // every hit counted, stored or not
            if (q == 0 && p.rowscale) p.rowscale[row0 + 16 * eg] = unscale[eg];
        }
}

// ---- dense score matrix on the same machinery: out[i][j] = A_i . B_j (svhip_score_matrix, the slab path's cohort GEMM) --------------------
// K = D = 192 / 256 is six / eight k steps: as a tiled GEMM (gemm_pw's split form, 256 x 128 tiles) the kernel spent its time fetching fp32
// operands (43 FLOP per byte) — 152 TFLOP/s, 0.68 ms for 16 384 x 16 384.  Here a wave keeps 32 rows of A in registers as half hi | lo parts
// and streams B (pre-split half planes, blocks of 32 rows through the LDS image of the AS-norm kernel) past them: three fp16 MFMAs per
// product block on v_mfma_f32_16x16x32_f16, scores stored straight from the accumulators (lane = 4 consecutive columns of one row: 16-byte
// stores, a 128-byte line per row per block).  Grid (row panels of 128, column slices): every workgroup walks `per` blocks of B.
#ifndef SCORE_ABL
#define SCORE_ABL 0          // tools only: 1 = stores straight from the accumulators (round 5's form)
#endif
template <int D>
__global__ __launch_bounds__(256, 2) void score_h3w_kernel(ScoreH3Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int CH = D / 8;
    constexpr int PL = 32 * D * 2;
    constexpr int BLK = 2 * PL;
    constexpr int NDMA = 2 * 32 * CH / 256;
    constexpr int NS = D / 32;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, q = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * 128 + wave * 32 + c;
    bool valid[2];
    int64_t arow[2];
#pragma unroll
    for (int eg = 0; eg < 2; ++eg) {
        arow[eg] = row0 + 16 * eg;
        valid[eg] = arow[eg] < p.Na;
    }
    bf16x8 bh[2][NS], bl[2][NS];
    float unscale[2] = {1.0f, 1.0f};            // per row of A: 1 / (its own scale * the scale of the B planes), exact powers of two
#pragma unroll
    for (int eg = 0; eg < 2; ++eg) {
        const float* src = p.A + (valid[eg] ? arow[eg] : p.Na - 1) * D;
        float sa = 1.0f;
        if (p.pscale) {                         // "operand scaling" above: the row by its own max |x|, the planes by the matrix's
            uint32_t m = 0;
#pragma unroll
            for (int s_ = 0; s_ < NS; ++s_) {
                const u32x4 w0 = *reinterpret_cast<const u32x4*>(src + 32 * s_ + 8 * q);
                const u32x4 w1 = *reinterpret_cast<const u32x4*>(src + 32 * s_ + 8 * q + 4);
#pragma unroll
                for (int u = 0; u < 4; ++u) m = max(m, max(finite_abs_bits(w0[u]), finite_abs_bits(w1[u])));
            }
            m = max(m, (uint32_t)__shfl_xor((int)m, 16, 64));
            m = max(m, (uint32_t)__shfl_xor((int)m, 32, 64));
            sa = pow2_scale_of_bits(m);
            unscale[eg] = pow2_inverse(sa) * pow2_inverse(pow2_scale_of_bits(*p.pscale));
        }
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(src + 32 * s_ + 8 * q);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(src + 32 * s_ + 8 * q + 4);
            f16x8 a8, b8;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float v = (u < 4 ? v0[u] : v1[u - 4]) * sa;
                const f16_t a = static_cast<f16_t>(v);
                a8[u] = a;
                b8[u] = static_cast<f16_t>(v - static_cast<float>(a));
            }
            bh[eg][s_] = __builtin_bit_cast(bf16x8, a8);
            bl[eg][s_] = __builtin_bit_cast(bf16x8, b8);
        }
    }
    const int nb_all = (p.Nb + 31) / 32;
    const int b_first = blockIdx.y * p.per, b_last = min(nb_all, b_first + p.per);
    if (b_first >= b_last) return;                                                  // workgroup-uniform
    const char* planes = reinterpret_cast<const char*>(p.planes);
    const int64_t plane_bytes = (int64_t)p.Nb * D * 2;
    auto issue = [&](int b, int buf) {
        const int r0 = b * 32;
        const int limit = p.Nb - 1 - r0;            // the last block clamps its rows
#pragma unroll
        for (int qq = 0; qq < NDMA; ++qq) {
            const int pidx = qq * 256 + tid;
            const int pl = pidx / (32 * CH), rem = pidx - pl * (32 * CH);
            const int i = rem / CH, cs = rem - i * CH;
            const int cc = (cs & ~7) | ((cs ^ (i >> 1)) & 7);
            const char* src = planes + pl * plane_bytes + ((int64_t)(r0 + min(i, limit)) * D + cc * 8) * 2;
            __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(smem + buf * BLK + (qq * 256 + wave * 64) * 16), 16, 0, 0);
        }
    };
    auto rd = [&](int buf, int pl, int rg, int s_) {
        const int i = 16 * rg + c, cs = 4 * s_ + q;
        return *reinterpret_cast<const bf16x8*>(smem + buf * BLK + pl * PL + i * (D * 2) + (((cs & ~7) | ((cs ^ (i >> 1)) & 7)) << 4));
    };
    const bool vec_ok = (p.ldo & 3) == 0 && (reinterpret_cast<uintptr_t>(p.out) & 15) == 0;
    // a wave whose 32 rows are all inside A issues exactly four 16-byte store instructions per interior block, AFTER the next block's DMAs:
    // the wait for block b may then leave those four (the newest entries of the in-order counter) in flight instead of draining them
    const bool rows_full = vec_ok && ((int64_t)blockIdx.x * 128 + wave * 32 + 32 <= p.Na);      // wave-uniform
    bool four_stores_behind = false;
    // Round 6: stored straight from the accumulators a wave instruction writes sixteen 64-byte row pieces (2.8 TB/s of
    // output at 16 384^2; plain stores of 256 contiguous bytes per row run at 6 TB/s: MI355X_MICROARCH.md).  Where the LDS has room beside
    // the two operand blocks (D = 192: 48 KiB + 4 x 8 KiB = 80 KiB, still two workgroups per CU) a wave keeps the scores of TWO blocks — 32
    // rows x 64 columns — in a private 8 KiB image (16-byte slot s of row r at s ^ (r & 15): conflict-free both ways) and writes them as
    // four rows x 256 contiguous bytes per instruction.  No barrier: the image is the wave's own, and a wave's LDS operations run in order.
    constexpr bool STAGED = D == 192 && !(SCORE_ABL & 1);
    char* stg = smem + 2 * BLK + wave * 8192;
    const bool stage = STAGED && vec_ok;
    int stores_behind = 0;                      // stores this wave issued behind the next block's DMAs (-1: not known)
    issue(b_first, 0);
    for (int b = b_first; b < b_last; ++b) {
        const int buf = (b - b_first) & 1;
        if (stage) {
            if (stores_behind == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (four_stores_behind) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // block b has landed; the previous block's stores may still fly
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        four_stores_behind = rows_full && (32 * b + 32 <= p.Nb);
        __syncthreads();
        if (b + 1 < b_last) issue(b + 1, buf ^ 1);
        f32x4 acc[2][2];
#pragma unroll
        for (int rg = 0; rg < 2; ++rg)
#pragma unroll
            for (int eg = 0; eg < 2; ++eg) acc[rg][eg] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 ah[2] = {rd(buf, 0, 0, 0), rd(buf, 0, 1, 0)}, al[2] = {rd(buf, 1, 0, 0), rd(buf, 1, 1, 0)};
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            bf16x8 nh[2] = {ah[0], ah[1]}, nl[2] = {al[0], al[1]};
            if (s_ + 1 < NS) {
#pragma unroll
                for (int rg = 0; rg < 2; ++rg) { nh[rg] = rd(buf, 0, rg, s_ + 1); nl[rg] = rd(buf, 1, rg, s_ + 1); }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rg = 0; rg < 2; ++rg)
#pragma unroll
                for (int eg = 0; eg < 2; ++eg) acc[rg][eg] = Half16<f16_t>::mfma16(al[rg], bh[eg][s_], acc[rg][eg]);
#pragma unroll
            for (int rg = 0; rg < 2; ++rg)
#pragma unroll
                for (int eg = 0; eg < 2; ++eg) acc[rg][eg] = Half16<f16_t>::mfma16(ah[rg], bl[eg][s_], acc[rg][eg]);
#pragma unroll
            for (int rg = 0; rg < 2; ++rg)
#pragma unroll
                for (int eg = 0; eg < 2; ++eg) acc[rg][eg] = Half16<f16_t>::mfma16(ah[rg], bh[eg][s_], acc[rg][eg]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rg = 0; rg < 2; ++rg) { ah[rg] = nh[rg]; al[rg] = nl[rg]; }
        }
        __builtin_amdgcn_s_setprio(0);
        // acc[rg][eg][e] = A row (16 eg + c of the wave) . B row 32 b + 16 rg + 4 q + e
        if (stage) {
            const int hb = (b - b_first) & 1;           // which half of the image this block fills
#pragma unroll
            for (int eg = 0; eg < 2; ++eg)
#pragma unroll
                for (int rg = 0; rg < 2; ++rg) {
                    const int slot = hb * 8 + rg * 4 + q;
                    *reinterpret_cast<f32x4*>(stg + (16 * eg + c) * 256 + ((slot ^ c) << 4)) = acc[rg][eg] * unscale[eg];
                }
            stores_behind = 0;
            if (hb == 1 || b + 1 == b_last) {
                asm volatile("" ::: "memory");
                const int jb = 32 * (b - hb);           // first column of the image
                const int t = lane & 15;
                const int64_t wrow0 = (int64_t)blockIdx.x * 128 + wave * 32;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int R = 4 * k + (lane >> 4);
                    const f32x4 o4 = *reinterpret_cast<const f32x4*>(stg + R * 256 + ((t ^ (R & 15)) << 4));
                    const int j0 = jb + 4 * t;
                    if (wrow0 + R < p.Na && t < 8 * (hb + 1)) {
                        float* orow = p.out + (wrow0 + R) * p.ldo;
                        if (j0 + 4 <= p.Nb) {
                            __builtin_nontemporal_store(o4, reinterpret_cast<f32x4*>(orow + j0));
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (j0 + e < p.Nb) orow[j0 + e] = o4[e];
                        }
                    }
                }
                // (exactly eight 16-byte store instructions when every row is inside A and both blocks are whole)
                stores_behind = (rows_full && hb == 1 && jb + 64 <= p.Nb) ? 8 : -1;
            }
            continue;
        }
#pragma unroll
        for (int eg = 0; eg < 2; ++eg) {
            if (!valid[eg]) continue;
            float* orow = p.out + arow[eg] * p.ldo;
#pragma unroll
            for (int rg = 0; rg < 2; ++rg) {
                const int j0 = 32 * b + 16 * rg + 4 * q;
                const f32x4 o4 = acc[rg][eg] * unscale[eg];
                if (vec_ok && j0 + 4 <= p.Nb) {
                    __builtin_nontemporal_store(o4, reinterpret_cast<f32x4*>(orow + j0));      // (written once, read by a later kernel at best)
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (j0 + e < p.Nb) orow[j0 + e] = o4[e];
                }
            }
        }
    }
}

// fp32 rows -> three bf16 planes [3][rows_total][D]: rows [0, n0) from A (the pseudo-cohort rows MB), the rest from B (the cohort)
__global__ __launch_bounds__(256) void split3_planes_kernel(const float* __restrict__ A, int n0, const float* __restrict__ B, int n1, int D,
                                                            bf16_t* __restrict__ planes) {
    const int64_t n = (int64_t)(n0 + n1) * D;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / D;
        const float v = r < n0 ? A[i] : B[i - (int64_t)n0 * D];
        const bf16_t a = static_cast<bf16_t>(v);
        const float r1 = v - static_cast<float>(a);
        const bf16_t b = static_cast<bf16_t>(r1);
        planes[i] = a;
        planes[n + i] = b;
        planes[2 * n + i] = static_cast<bf16_t>(r1 - static_cast<float>(b));
    }
}
// ... -> two half planes [2][rows_total][D] (hi, lo) for the three-fp16-MFMA form
// `pscale` (optional): the max-|x| word of B (absmax_bits_kernel): every row is multiplied by s = pow2_scale_of_bits(*pscale), the first
// `sq_rows` rows of A (second-moment rows: products of two cohort values) by s^2
__global__ __launch_bounds__(256) void split2_planes_kernel(const float* __restrict__ A, int n0, const float* __restrict__ B, int n1, int D,
                                                            f16_t* __restrict__ planes, uint32_t* __restrict__ pscale, int sq_rows) {
    const int64_t n = (int64_t)(n0 + n1) * D;
    float sc = 1.0f;
    if (pscale) {           // pscale[1 ..] = the partials of absmax_bits_kernel; pscale[0] <- their maximum, for the kernels behind this one
        const uint32_t bits = absmax_fold(pscale + 1);
        if (blockIdx.x == 0 && threadIdx.x == 0) *pscale = bits;
        sc = pow2_scale_of_bits(bits);
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / D;
        const float v = (r < n0 ? A[i] : B[i - (int64_t)n0 * D]) * (r < sq_rows ? sc * sc : sc);
        const f16_t a = static_cast<f16_t>(v);
        planes[i] = a;
        planes[n + i] = static_cast<f16_t>(v - static_cast<float>(a));
    }
}

// one wave per embedding: exact statistics of the top-`top` of its candidates; rows that cannot be decided are flagged.
// NL candidate lists per embedding (2: one per half-wave lane of the 32-wide kernels; 4: the 16-wide kernel), 2 * AF_CAPL slots in all.
template <int NL>
__global__ __launch_bounds__(256) void asnorm_cand_stats_kernel(const float* __restrict__ cand, const int32_t* __restrict__ cnt, int64_t rows,
                                                                int top, float* __restrict__ mu, float* __restrict__ sigma,
                                                                int64_t row_base, int32_t* __restrict__ flagged, int32_t* __restrict__ nflag,
                                                                const float* __restrict__ rowscale, int32_t* __restrict__ finfo) {
    constexpr int CAP = 2 * AF_CAPL / NL;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    // every slot of the row is loaded at once, BEFORE the counts are known — one memory round trip instead of two (this kernel is one wave per
    // embedding and lives on latency); a slot past its list's count becomes key 0 below.  (Never-written slots hold whatever the scratch
    // held: they are masked by the counts.)
    const float* c = cand + row * 2 * AF_CAPL;
    uint32_t ck[2 * AF_CAPL / 64];
    float cv[2 * AF_CAPL / 64];
#pragma unroll
    for (int q = 0; q < 2 * AF_CAPL / 64; ++q) cv[q] = __builtin_nontemporal_load(c + lane + 64 * q);
    int cn[NL], total = 0;
    bool over = false;
#pragma unroll
    for (int l = 0; l < NL; ++l) { cn[l] = cnt[row * NL + l]; total += cn[l]; over |= cn[l] > CAP; }
    if (total < top || over) {
        if (lane == 0) {
            const int slot = atomicAdd(nflag, 1);
            flagged[slot] = (int32_t)(row_base + row);
            if (finfo) finfo[slot] = min(total, (1 << 30) - 1) | (over ? (1 << 30) : 0);       // what the refit pass starts from
        }
        return;
    }
    int nq = 2 * AF_CAPL / 64;
    if (NL == 4) {
        // the 16-wide kernel interleaves its four lists: float p of the region = element p >> 2 of list p & 3 — dense up to 4 * (shortest
        // list), empty past 4 * (longest): the search registers past that are skipped (typically 5 of 8)
        const int cl = (lane & 3) == 0 ? cn[0] : (lane & 3) == 1 ? cn[1] : (lane & 3) == 2 ? cn[2 % NL] : cn[3 % NL];
        int cmax = 0;
#pragma unroll
        for (int k = 0; k < NL; ++k) cmax = max(cmax, cn[k]);
        nq = min(nq, __builtin_amdgcn_readfirstlane((4 * cmax + 63) >> 6));
#pragma unroll
        for (int q = 0; q < 2 * AF_CAPL / 64; ++q) ck[q] = ((lane >> 2) + 16 * q) < cl ? fkey(cv[q]) : 0u;
    } else {
#pragma unroll
        for (int q = 0; q < 2 * AF_CAPL / 64; ++q) {
            const int pos = lane + 64 * q, l = pos / CAP, idx = pos - l * CAP;
            int cl = 0;
#pragma unroll
            for (int k = 0; k < NL; ++k) cl = l == k ? cn[k] : cl;
            ck[q] = idx < cl ? fkey(cv[q]) : 0u;
        }
    }
    float m, sd;
    // (total >= top populated keys was checked above: the search may start below the bits they share.  The register count of the search is a
    //  compile-time constant per case: with a run-time bound inside select_stats the eight ballot chains of a step sit behind eight uniform
    //  branches and no longer interleave — the guarded form measured 25 % SLOWER than searching all eight registers)
    auto run = [&](auto nreg) {
        constexpr int NR = decltype(nreg)::value;
        uint32_t c[NR];
#pragma unroll
        for (int j = 0; j < NR; ++j) c[j] = ck[j];
        select_stats<NR>(c, top, m, sd, NR, true);
    };
    if (nq <= 5) run(std::integral_constant<int, 5>{});
    else if (nq == 6) run(std::integral_constant<int, 6>{});
    else run(std::integral_constant<int, 2 * AF_CAPL / 64>{});
    // (candidates of the scaled kernels are stored in the scaled domain: mean and deviation are linear in the scale, a power of two)
    const float rs = rowscale ? rowscale[row] : 1.0f;
    if (lane == 0) { mu[row_base + row] = m * rs; sigma[row_base + row] = sd * rs; }
}

// MB = [M ; cbar ; 0]: M[i][j] = (1/K) sum_k C[k][i] C[k][j] (D x D), cbar[j] = (1/K) sum_k C[k][j].
// Stage 1: grid (D / 32, D / 32 + 1, MOM_SLICES): a 32 x 32 tile of C^T C over one slice of the cohort rows (thread = column j, 4 rows i;
// the extra block row sums the columns themselves); stage 2 adds the slices in a fixed order (no atomics: same bits every run).
constexpr int MOM_SLICES = 16;
__global__ __launch_bounds__(256) void cohort_moments_part_kernel(const float* __restrict__ C, int K, int D, float* __restrict__ part) {
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int j = blockIdx.x * 32 + tx;
    const int per = (K + MOM_SLICES - 1) / MOM_SLICES;
    const int k0 = blockIdx.z * per, k1 = min(K, k0 + per);
    float* out = part + (int64_t)blockIdx.z * (D + 32) * D;
    if (blockIdx.y == gridDim.y - 1) {          // cbar (row D), zeros below
        float s = 0.0f;
        if (ty == 0) for (int k = k0; k < k1; ++k) s += C[(int64_t)k * D + j];
        for (int r = ty; r < 32; r += 8) out[(int64_t)(D + r) * D + j] = r == 0 ? s : 0.0f;
        return;
    }
    const int i0 = blockIdx.y * 32 + ty * 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = k0; k < k1; ++k) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(C + (int64_t)k * D + i0);
        const float b = C[(int64_t)k * D + j];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = fmaf(a[u], b, acc[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) out[(int64_t)(i0 + u) * D + j] = acc[u];
}
__global__ __launch_bounds__(256) void cohort_moments_sum_kernel(const float* __restrict__ part, int n, float inv_k, float* __restrict__ MB) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 0.0f;
#pragma unroll
    for (int z = 0; z < MOM_SLICES; ++z) s += part[(int64_t)z * n + i];
    MB[i] = s * inv_k;
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ E, const int32_t* __restrict__ ids, int n, int D,
                                                          float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)n * D) return;
    const int r = (int)(i / D), k = (int)(i - (int64_t)r * D);
    out[i] = E[(int64_t)ids[r] * D + k];
}
__global__ __launch_bounds__(256) void scatter_stats_kernel(const float* __restrict__ m, const float* __restrict__ s, const int32_t* __restrict__ ids,
                                                            int n, float* __restrict__ mu, float* __restrict__ sigma) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    mu[ids[i]] = m[i];
    sigma[ids[i]] = s[i];
}

template <int D>
hipError_t launch_fused_d(const AsnormFusedParams& p, hipStream_t stream) {
    static DeviceOnce attr;
    const int lds = 2 * 32 * D * 4;
    if (hipError_t e = set_max_dynamic_lds(attr, reinterpret_cast<const void*>(asnorm_fused_kernel<D>), lds)) return e;
    hipLaunchKernelGGL((asnorm_fused_kernel<D>), dim3((unsigned)((p.N + 127) / 128)), dim3(256), lds, stream, p);
    return hipGetLastError();
}

}  // namespace

bool asnorm_fused_supported(int D, int K, int top) {
    return (D == 192 || D == 256) && top >= 1 && top <= 256 && K >= 4 * top && K >= 64;
}

// Upper-tail normal quantile z with P(x > mean + z sd) = frac (Abramowitz-Stegun 26.2.22), frac <= 0.5
float asnorm_tail_z(int K, int top) {
    const float frac = fminf(0.5f, 1.6f * (float)top / (float)K);
    const float tq = sqrtf(-2.0f * logf(fmaxf(frac, 1e-6f)));
    return tq - (2.30753f + 0.27061f * tq) / (1.0f + 0.99229f * tq + 0.04481f * tq * tq);
}

size_t cohort_moments_scratch_bytes(int D) { return (size_t)MOM_SLICES * (D + 32) * D * sizeof(float); }

hipError_t launch_cohort_moments(const float* cohort, int K, int D, float* MB, float* part, hipStream_t stream) {
    if (D % 32 != 0 || K <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(cohort_moments_part_kernel, dim3(D / 32, D / 32 + 1, MOM_SLICES), dim3(256), 0, stream, cohort, K, D, part);
    const int n = (D + 32) * D;
    hipLaunchKernelGGL(cohort_moments_sum_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, part, n, 1.0f / (float)K, MB);
    return hipGetLastError();
}

// planes = 2: two half planes, three fp16 MFMAs (D = 192 / 256); planes = 3: three bf16 planes, six bf16 MFMAs (D = 192)
bool asnorm_fused6_supported(int D, int planes) { return planes == 2 ? (D == 192 || D == 256) : (planes == 3 && D == 192); }
size_t asnorm_planes_bytes(int D, int K) { return (size_t)3 * (D + 32 + K) * D * 2; }

hipError_t launch_asnorm_planes(const float* MB, const float* cohort, int K, int D, void* planes, hipStream_t stream, int nplanes, uint32_t* pscale) {
    if (!MB || !cohort || !planes || K <= 0 || D % 32 != 0 || !(nplanes == 2 || nplanes == 3) || (pscale && nplanes != 2)) return hipErrorInvalidValue;
    const int64_t n = (int64_t)(D + 32 + K) * D;
    const int64_t g = (n + 255) / 256;
    if (pscale)             // the cohort's max |x| -> pscale[1 ..] (partials; the split pass folds them into pscale[0]); cohort rows and the mean row are scaled by s, the D moment rows by s^2
        hipLaunchKernelGGL(absmax_bits_kernel, dim3(ABSMAX_WGS), dim3(256), 0, stream, cohort, (int64_t)K * D, pscale + 1);
    if (nplanes == 2) hipLaunchKernelGGL(split2_planes_kernel, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, stream, MB, D + 32, cohort, K, D, reinterpret_cast<f16_t*>(planes), pscale, D);
    else hipLaunchKernelGGL(split3_planes_kernel, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, stream, MB, D + 32, cohort, K, D, reinterpret_cast<bf16_t*>(planes));
    return hipGetLastError();
}

hipError_t launch_asnorm_fused(const AsnormFusedParams& p, int D, hipStream_t stream) {
    if (p.N <= 0) return hipSuccess;
    if (!asnorm_fused_supported(D, p.K, 1) || !p.E || !p.cohort || !p.MB || !p.cand || !p.cnt) return hipErrorInvalidValue;
    if ((reinterpret_cast<uintptr_t>(p.E) | reinterpret_cast<uintptr_t>(p.cohort) | reinterpret_cast<uintptr_t>(p.MB)) & 15) return hipErrorInvalidValue;
    if (p.nlists != 2 && !(p.nlists == 4 && p.planes && p.nplanes == 2)) return hipErrorInvalidValue;      // (four lists: the 16-wide half-plane kernel only)
    if (p.planes) {                                       // the split forms: three fp16 MFMAs on two half planes / six bf16 MFMAs on three
        if (!asnorm_fused6_supported(D, p.nplanes) || (reinterpret_cast<uintptr_t>(p.planes) & 15) || p.N > (int64_t)1 << 21) return hipErrorInvalidValue;      // (32-bit lane offsets)
        if (p.nlists == 4 && p.N * 2 * ASNORM_CAND_PER_LANE * 4 >= (int64_t)1 << 32) return hipErrorInvalidValue;      // (32-bit byte offsets into the candidate lists)
        if (p.pscale && (p.nlists != 4 || !p.rowscale)) return hipErrorInvalidValue;                                    // (scaled planes: the 16-wide kernel, with its row factors)
        const dim3 grid((unsigned)((p.N + 127) / 128));
#define SV_AF6(DD, NP)                                                                                                      \
        {                                                                                                                   \
            static DeviceOnce attr6;                                                                                        \
            constexpr int lds6 = 2 * NP * 32 * DD * 2;                                                                      \
            if (hipError_t e = set_max_dynamic_lds(attr6, reinterpret_cast<const void*>(asnorm_fused6_kernel<DD, NP>), lds6)) return e; \
            hipLaunchKernelGGL((asnorm_fused6_kernel<DD, NP>), grid, dim3(256), lds6, stream, p);                           \
            return hipGetLastError();                                                                                       \
        }
        if (p.nplanes == 3) SV_AF6(192, 3)
        if (p.nlists == 4) {            // two half planes on the 16-wide MFMA (the default)
#define SV_H3W(DD)                                                                                                          \
            {                                                                                                               \
                static DeviceOnce attrw;                                                                                    \
                constexpr int ldsw = 2 * 2 * 32 * DD * 2;                                                                   \
                if (hipError_t e = set_max_dynamic_lds(attrw, reinterpret_cast<const void*>(asnorm_h3w_kernel<DD>), ldsw)) return e; \
                hipLaunchKernelGGL((asnorm_h3w_kernel<DD>), grid, dim3(256), ldsw, stream, p);                              \
                return hipGetLastError();                                                                                   \
            }
            if (D == 192) SV_H3W(192)
            SV_H3W(256)
#undef SV_H3W
        }
        if (D == 192) SV_AF6(192, 2)
        SV_AF6(256, 2)
#undef SV_AF6
    }
    switch (D) {
        case 192: return launch_fused_d<192>(p, stream);
        case 256: return launch_fused_d<256>(p, stream);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_asnorm_cand_stats(const float* cand, const int32_t* cnt, int64_t rows, int top, float* mu, float* sigma, int64_t row_base,
                                    int32_t* flagged, int32_t* nflag, hipStream_t stream, int nlists, const float* rowscale, int32_t* finfo) {
    if (rows <= 0) return hipSuccess;
    if (nlists == 4)
        hipLaunchKernelGGL(asnorm_cand_stats_kernel<4>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, cand, cnt, rows, top, mu, sigma, row_base,
                           flagged, nflag, rowscale, finfo);
    else if (nlists == 2)
        hipLaunchKernelGGL(asnorm_cand_stats_kernel<2>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, cand, cnt, rows, top, mu, sigma, row_base,
                           flagged, nflag, rowscale, finfo);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// ---- refit state of the embeddings the normal-quantile threshold did not fit (round 6) -------------------------------------------------
// Structure of arrays, stride = the number of rows: [id (int bits) | z | z of the pass before | ln(count) of the pass before | zlo | zhi]
// (zlo: the largest z that passed too many, zhi: the smallest that passed too few; <= -50: none yet).  A row's next z from what its last pass
// counted: ln(count) against z is locally close to linear — slope ~ -(z + 1 / z) for a Gaussian tail, several times steeper when the
// threshold sits on the flank of a narrow mode (clustered speaker centroids) — so the FIRST step assumes a slope of -5 and every later step
// is the secant through the last two measurements, kept inside the bracket once both sides have been seen.
__device__ __forceinline__ void refit_next_z(float& z, float& zp, float& lcp, float& zlo, float& zhi, int32_t inf, float target) {
    const float lnT = __logf(target);
    const float c = (float)(inf & ((1 << 30) - 1));
    const bool many = ((inf >> 30) & 1) || c > target;       // (a flagged row with enough candidates overflowed a list)
    const float lc = __logf(fmaxf(many ? fmaxf(c, 1.25f * target) : c, 0.5f));
    if (many) zlo = zlo > -50.f ? fmaxf(zlo, z) : z; else zhi = zhi > -50.f ? fminf(zhi, z) : z;
    float slope = -5.0f;
    if (zp > -50.f && fabsf(z - zp) > 1e-3f) slope = fminf(-1.0f, fmaxf(-12.0f, (lc - lcp) / (z - zp)));
    float zn = z + fminf(0.6f, fmaxf(-0.6f, (lnT - lc) / slope));
    if (many) zn = fmaxf(zn, z + 0.02f); else zn = fminf(zn, z - 0.02f);
    if (zlo > -50.f && zhi > -50.f) {
        const float w = zhi - zlo;
        zn = fminf(zhi - 0.1f * w, fmaxf(zlo + 0.1f * w, zn));
    }
    zp = z; lcp = lc;
    z = fminf(12.0f, fmaxf(-2.0f, zn));
}
__global__ __launch_bounds__(256) void refit_init_kernel(const int32_t* __restrict__ ids, const int32_t* __restrict__ info, int n, float z0, float target,
                                                         float* __restrict__ soa) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float z = z0, zp = -100.f, lcp = 0.f, zlo = -100.f, zhi = -100.f;
    refit_next_z(z, zp, lcp, zlo, zhi, info[i], target);
    reinterpret_cast<int32_t*>(soa)[i] = ids[i];
    soa[n + i] = z; soa[2 * (size_t)n + i] = zp; soa[3 * (size_t)n + i] = lcp; soa[4 * (size_t)n + i] = zlo; soa[5 * (size_t)n + i] = zhi;
}
__global__ __launch_bounds__(256) void refit_next_kernel(const float* __restrict__ in, int n_in, const int32_t* __restrict__ pos, const int32_t* __restrict__ info,
                                                         int left, float target, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= left) return;
    const int j = pos[i];
    float z = in[n_in + j], zp = in[2 * (size_t)n_in + j], lcp = in[3 * (size_t)n_in + j], zlo = in[4 * (size_t)n_in + j], zhi = in[5 * (size_t)n_in + j];
    refit_next_z(z, zp, lcp, zlo, zhi, info[i], target);
    reinterpret_cast<int32_t*>(out)[i] = reinterpret_cast<const int32_t*>(in)[j];
    out[left + i] = z; out[2 * (size_t)left + i] = zp; out[3 * (size_t)left + i] = lcp; out[4 * (size_t)left + i] = zlo; out[5 * (size_t)left + i] = zhi;
}
hipError_t launch_asnorm_refit_init(const int32_t* ids, const int32_t* info, int n, float z0, float target, float* soa, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(refit_init_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, ids, info, n, z0, target, soa);
    return hipGetLastError();
}
hipError_t launch_asnorm_refit_next(const float* in, int n_in, const int32_t* pos, const int32_t* info, int left, float target, float* out, hipStream_t stream) {
    if (left <= 0) return hipSuccess;
    hipLaunchKernelGGL(refit_next_kernel, dim3((left + 255) / 256), dim3(256), 0, stream, in, n_in, pos, info, left, target, out);
    return hipGetLastError();
}

hipError_t launch_gather_rows(const float* E, const int32_t* ids, int n, int D, float* out, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)(((int64_t)n * D + 255) / 256)), dim3(256), 0, stream, E, ids, n, D, out);
    return hipGetLastError();
}

hipError_t launch_scatter_stats(const float* m, const float* s, const int32_t* ids, int n, float* mu, float* sigma, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(scatter_stats_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, m, s, ids, n, mu, sigma);
    return hipGetLastError();
}

bool score_h3w_supported(int D, int64_t Na, int64_t Nb) { return (D == 192 || D == 256) && Na > 0 && Nb > 0 && Nb < ((int64_t)1 << 30) && Na < ((int64_t)1 << 31); }
size_t score_h3w_planes_bytes(int D, int64_t Nb) { return (size_t)2 * Nb * D * 2 + 2048; }      // two half planes + the scale word of B and its 256 partials behind them

// out (Na, ldo) = A (Na, D) . B (Nb, D)^T; `planes` = score_h3w_planes_bytes of scratch (filled here with the half parts of B)
hipError_t launch_score_h3w(const float* A, int64_t Na, const float* B, int64_t Nb, int D, float* out, int64_t ldo, void* planes, int num_cu,
                            hipStream_t stream) {
    if (!score_h3w_supported(D, Na, Nb) || !A || !B || !out || !planes || ldo < Nb) return hipErrorInvalidValue;
    if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) | reinterpret_cast<uintptr_t>(planes)) & 15) return hipErrorInvalidValue;
    const int64_t n = Nb * D;
    const int64_t g = (n + 255) / 256;
    uint32_t* pscale = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(planes) + (size_t)2 * Nb * D * 2);      // (16-byte aligned: Nb * D * 4 bytes in)
    hipLaunchKernelGGL(absmax_bits_kernel, dim3(ABSMAX_WGS), dim3(256), 0, stream, B, n, pscale + 1);
    hipLaunchKernelGGL(split2_planes_kernel, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, stream, B, 0, B, (int)Nb, D, reinterpret_cast<f16_t*>(planes), pscale, 0);
    ScoreH3Params p;
    p.A = A; p.Na = Na; p.planes = planes; p.Nb = (int)Nb; p.out = out; p.ldo = ldo; p.pscale = pscale;
    const int panels = (int)((Na + 127) / 128), nb_all = (int)((Nb + 31) / 32);
    // column slices: enough workgroups for ~4 per CU, at least 8 blocks per slice
    int slices = (4 * (num_cu > 0 ? num_cu : 256) + panels - 1) / panels;
    if (slices > (nb_all + 7) / 8) slices = (nb_all + 7) / 8;
    if (slices < 1) slices = 1;
    if (slices > 65535) slices = 65535;
    p.per = (nb_all + slices - 1) / slices;
    slices = (nb_all + p.per - 1) / p.per;
    const dim3 grid((unsigned)panels, (unsigned)slices);
#define SV_SC(DD)                                                                                                           \
    {                                                                                                                       \
        static DeviceOnce attr;                                                                                             \
        constexpr int lds = 2 * 2 * 32 * DD * 2 + (DD == 192 ? 4 * 8192 : 0);     /* + the waves' store images (score_h3w_kernel) */ \
        if (hipError_t e = set_max_dynamic_lds(attr, reinterpret_cast<const void*>(score_h3w_kernel<DD>), lds)) return e;    \
        hipLaunchKernelGGL((score_h3w_kernel<DD>), grid, dim3(256), lds, stream, p);                                        \
        return hipGetLastError();                                                                                           \
    }
    if (D == 192) SV_SC(192)
    SV_SC(256)
#undef SV_SC
}

}  // namespace svhip
