// rn_block128.hip — one RawNetBasicBlock with 128 -> 128 channels and a max-pool (RawNet2 layer1 / layer2) as ONE kernel (bf16).
//
// Reference: models/RawNet_baseline.py:221-232 (RawNetBasicBlock.forward), :62-68 (AFMS of the PREVIOUS block, applied on
// the way in), models/RawNet2_custom.py:230-243 (layers = [1, 1, ...]: both 128-channel stages are a single pooled block).
//
//   y    = (xin + alpha) * gate[b]            (the previous block's AFMS; block 0: y = xin)
//   pre  = lrelu0.3(bn1(y))                                                                     :222
//   h    = lrelu0.3(bn2(conv1_k3_p1(pre)))                                                      :224-225
//   o    = conv2_k3_p1(h) + y                 (identity shortcut: the pre-BN input)             :223,226-227
//   out  = max_pool1d(o, 3)                                                                     :228-229
//   colsum[b, part, c] = partial sums over pooled frames of out   (feeds THIS block's AFMS mean, :64)
//
// These two blocks are 41 % of the model's FLOPs on the longest time axes (T = 10583 / 3527 frames): as separate launches
// they moved ~7 GB per batch through HBM (pre-activation, h, o, pool, mean, gate: 2.5 ms of a 5.2 ms step); fused, an
// utterance tile is read once and only the pooled output (1/3 of the frames) is written.
//
// Structure (gfx950):
//   * persistent workgroups, one per CU, 512 threads = 8 waves = two wave groups with fixed roles, one wave of each per SIMD:
//       group A (waves 0-3): input conversion (gate, BN, LeakyReLU) and conv1;   group B (waves 4-7): conv2, shortcut, pool.
//     Wave w of a group owns output channels 32w .. 32w+31 of ITS convolution; the weight fragments (2 x 12 x 16 B per lane =
//     96 VGPRs) stay in registers for the whole kernel as the MFMA A operand (v_mfma_f32_16x16x32_bf16): weights are read
//     once per workgroup, and a lane owns 4 consecutive channels of one frame in the accumulators.
//   * work item = (utterance, tile of 78 output frames = 26 pooled frames).  The groups run ONE ITEM APART in rounds of two
//     phases (two workgroup barriers per round):
//         matrix phase:  A: conv1 of item r (P -> H[r & 1])        |  B: conv2 of item r-1 (H, Y -> O)
//         vector phase:  A: converts item r+1 in place (RAW -> P, Y), requests item r+2  |  B: pools item r-1 (O -> global), column sums
//     so each SIMD has TWO matrix streams to interleave in the matrix phase (one wave's fragment reads and epilogue issue in
//     the shadow of the other's MFMAs) and two vector streams in the vector phase.  (First version: A's matrix phase beside
//     B's vector phase and vice versa — a lone matrix stream per SIMD cannot hide its own LDS-read issue: 0.62 ms -> this.)
//   * activation tiles live in LDS, frame-major, 256-byte rows, 16-byte chunk c of row r at chunk c ^ RB_SWZ(r): a k = 3
//     convolution is three shifted reads of the same rows — no im2col; 2 + 2 halo frames per tile are recomputed (2.5 %).
//   * input rows travel global -> LDS by DMA (the swizzle applied on the per-lane SOURCE chunk, as the DMA writes LDS lane-linear)
//     into one of two P buffers, two rounds before conv1 needs them; the conversion rewrites them in place.  No load latency
//     is exposed and no registers are held across the matrix phase.
#include "common.h"
#include "kernels.h"

namespace svhip {

namespace {

constexpr int RB_TT = 78;                     // output frames per tile (multiple of 3; RB_TT + 2 = 5 MFMA frame blocks)
constexpr int RB_FB = 5;
constexpr int RB_PR = 82, RB_HR = 82, RB_YR = 80, RB_OR = 80;
constexpr int RB_RAWK = 21;                   // 1 KiB DMA pieces per item (4 rows each, 82 rows)
constexpr int RB_P = 0;                                          // 2 x 21 KiB: DMA target, converted in place, conv1's operand
constexpr int RB_H = RB_P + 2 * RB_RAWK * 1024;                  // 2 x 82 rows
constexpr int RB_Y = RB_H + 2 * RB_HR * 256;                     // 2 x 80 rows
constexpr int RB_O = RB_Y + 2 * RB_YR * 256;
constexpr int RB_CST = RB_O + RB_OR * 256;                       // 5 x 128 floats of per-channel constants
constexpr int RB_GATE = RB_CST + 5 * 128 * 4;                    // 2 x 1 KiB: the item's AFMS gate row (one DMA piece: the 128 floats twice)
constexpr int RB_LDS = RB_GATE + 2 * 1024;                       // 151 040 bytes

typedef bf16_t bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // LDS traffic only: global loads / stores in flight survive the barrier
    __builtin_amdgcn_s_barrier();
}

__device__ __forceinline__ float lrelu03(float t) { return fmaxf(t, 0.3f * t); }

// LDS tiles: 256-byte rows; 16-byte chunk c of row r lives at chunk c ^ RB_SWZ(r).  Found by exhaustive search over the GF(2)-linear
// maps of the row bits: (r & 7) << 1 is conflict-free for the ds_read_b128 lane groups of the 16x16x32 B-operand fetch (16
// consecutive rows x two adjacent k chunks) for EVERY tap shift of the rows; r & 15 is not (2-way on the odd tap).  The 8-byte
// accumulator-layout accesses and the row-per-16-lanes elementwise passes are 2-way under any such map; they are 1/6 of the traffic.
#define RB_SWZ(r) (((r) & 7) << 1)
template <int N> struct RbInt { static constexpr int value = N; };

// two 16-bit values in one dword: unpack to fp32 (exact), pack with round-to-nearest-even — Half16<H>::lo / hi / pack2 (common.h),
// H = bf16_t or f16_t (the kernel moves raw 16-byte chunks; only these conversions and the MFMA opcode know the type)
#define bf_lo Half16<H>::lo
#define bf_hi Half16<H>::hi
#define bf_pack Half16<H>::pack2

// The two wave groups run disjoint code in one loop; whatever the compiler hoists out of that loop (LDS addresses, per-lane
// constants of BOTH roles) stays live across everything, pushed this kernel over its 256 registers, and every use of a spilled
// value (scratch = vector memory) then waited for ALL outstanding DMAs and stores: 2x on the whole kernel.  Lane coordinates
// made opaque inside a phase are recomputed there (a few VALU ops) and die at its end.
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }

// out[ch][frame] += W[ch][tap*128 + c] * src[frame + tap][c] for 5 frame blocks x this wave's 32 channels.  K runs in 12 steps
// of 32 (tap, 32 channels); the activation fragments of step s + 1 are requested while the 10 MFMAs of step s issue.
// INIT: the accumulators start from a bf16 tile (the identity shortcut of conv2: o = y + conv2(h)) instead of zero.
template <bool INIT, typename H>
__device__ __forceinline__ void conv_k3(const char* src, const bf16x8 (&w)[2][12], f32x4 (&acc)[RB_FB][2], int r16, int q4,
                                        const char* init, int w4, int dbg = 0) {
    bf16x8 xb[3][RB_FB];
#ifdef SVHIP_GEMM_DEBUG
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int fb = 0; fb < RB_FB; ++fb) { u32x4 z = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}; __builtin_memcpy(&xb[i][fb], &z, 16); }
#endif
    // row = fb*16 + r16 + tap: the swizzle key (row & 15) does not depend on fb, so one base per (tap, kc) + an immediate fb offset
    auto fetch = [&](int ks, bf16x8 (&dst)[RB_FB]) {
        const int tap = ks >> 2, kc = ks & 3;
        const int rt = r16 + tap;
        const char* base = src + rt * 256 + (((kc * 4 + q4) ^ RB_SWZ(rt)) << 4);
#pragma unroll
        for (int fb = 0; fb < RB_FB; ++fb) dst[fb] = *reinterpret_cast<const bf16x8*>(base + fb * 4096);
    };
#ifdef SVHIP_GEMM_DEBUG
    const bool no_rd = dbg & 1, no_mm = dbg & 2;
#else
    constexpr bool no_rd = false, no_mm = false;
#endif
    if (!no_rd) { fetch(0, xb[0]); fetch(1, xb[1]); }
    __builtin_amdgcn_s_setprio(1);              // the SIMD's other wave is in its vector / LDS phase: the matrix stream wins issue arbitration
    if (INIT) {
        // A lane's 4 channels of one row are 8 bytes; rows 16 apart share a swizzle key, so 8-byte accesses of 16 rows are 2-way
        // bank conflicts.  Lanes l and l + 16 (q4 even / odd) own the two halves of one 16-byte chunk: the even one fetches the
        // whole chunk of channel block 0, the odd one that of block 1, and v_permlane16_swap hands each its own halves.
#pragma unroll
        for (int fb = 0; fb < RB_FB; ++fb) {
            const int q = fb * 16 + r16, c16 = w4 * 4 + (q4 & 1) * 2 + (q4 >> 1);
            const u32x4 y4 = *reinterpret_cast<const u32x4*>(init + q * 256 + ((c16 ^ RB_SWZ(q)) << 4));
            const auto s0 = __builtin_amdgcn_permlane16_swap(y4[0], y4[2], false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(y4[1], y4[3], false, false);
            acc[fb][0] = f32x4{bf_lo(s0[0]), bf_hi(s0[0]), bf_lo(s1[0]), bf_hi(s1[0])};
            acc[fb][1] = f32x4{bf_lo(s0[1]), bf_hi(s0[1]), bf_lo(s1[1]), bf_hi(s1[1])};
        }
    } else {
#pragma unroll
        for (int fb = 0; fb < RB_FB; ++fb)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) acc[fb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int ks = 0; ks < 12; ++ks) {
        // fragments are requested TWO K steps (20 MFMAs = 320 cycles) ahead of their use; the scheduler may not move anything
        // across these fences: left alone it sinks each read next to its first use and the wave then waits a full LDS round trip
        // in front of every MFMA pair (measured: 3x the MFMA time)
        if (ks + 2 < 12 && !no_rd) fetch(ks + 2, xb[(ks + 2) % 3]);
        if (!no_mm)
#pragma unroll
        for (int fb = 0; fb < RB_FB; ++fb)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
                acc[fb][cb] = Half16<H>::mfma16(w[cb][ks], xb[ks % 3][fb], acc[fb][cb]);
        // inside the fenced region: two MFMAs, then one of the reads (for K step ks + 2) in their shadow — issued back to back the five
        // reads hold the wave's issue port for ~65 cycles per step that no MFMA covers (measured: +0.8k cycles per convolution)
#pragma unroll
        for (int g = 0; g < RB_FB; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(0);
}

template <bool GATE, typename H>
__global__ __launch_bounds__(512, 2) void rn_block128_kernel(RnBlock128Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, w4 = wave & 3;                     // group A = 0 (conversion + conv1), B = 1 (conv2 + pool)
    const int r16_ = lane & 15, q4_ = lane >> 4;
    const int gt_ = tid & 255;                                    // thread index inside the group
    const int c16_ = gt_ & 15, rg_ = gt_ >> 4;                    // elementwise passes: 16-byte chunk (channels 8*c16 ..), row group

    // ---- this wave's weights: output channels w4*32 + cb*16 + r16, k = ks*32 + 8*q4 .. +7 (k = tap*128 + c) ----
    bf16x8 wf[2][12];
    {
        const bf16_t* W = grp == 0 ? p.W1 : p.W2;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int ks = 0; ks < 12; ++ks)
                wf[cb][ks] = *reinterpret_cast<const bf16x8*>(W + (int64_t)(w4 * 32 + cb * 16 + r16_) * 384 + ks * 32 + q4_ * 8);
    }
    // per-channel constants in LDS: [bn1 scale | bn1 shift | alpha | bn2 scale | bn2 shift] x 128 floats.  (No ordinary global
    // load may sit in the steady-state loop: hipcc drains every LDS-DMA in flight, vmcnt(0), at the first use of its result.)
    for (int i = tid; i < 5 * 128; i += 512) {
        const int which = i >> 7, c = i & 127;
        const float* srcs[5] = {p.bn1_scale, p.bn1_shift, p.alpha, p.bn2_scale, p.bn2_shift};
        reinterpret_cast<float*>(smem + RB_CST)[i] = (which == 2 && !GATE) ? 0.0f : srcs[which][c];
    }
    for (int i = tid; i < 2 * 2 * 16; i += 512)                   // H rows 80, 81 (both buffers) feed only the two discarded output rows
        *reinterpret_cast<u32x4*>(smem + RB_H + (i >> 5) * RB_HR * 256 + (80 + ((i >> 4) & 1)) * 256 + ((i & 15) << 4)) = u32x4{0u, 0u, 0u, 0u};

    // this workgroup's items: the contiguous range [first, first + n_mine) of (utterance, tile) pairs — consecutive tiles of ONE
    // utterance (rarely two), so the column sums of the pooled output accumulate in registers and leave once per utterance
    const int items = p.B * p.ntiles;
    const int first = (int)blockIdx.x * p.per_wg;
    const int n_mine = max(0, min(p.per_wg, items - first));

    // group A: wave-instruction `piece` moves rows 4*piece .. +3 of the item's 82 input rows (1 KiB, lane-linear in LDS) into P[k & 1];
    // lane l fills LDS chunk (l & 15) of its row, which holds DATA chunk (l & 15) ^ RB_SWZ(row).  Frames outside [0, T) read a
    // clamped row and are zeroed by the conversion.
    auto issue_dma = [&](int k) {
        const int item = first + k;
        const int b = item / p.ntiles, t0 = (item - b * p.ntiles) * RB_TT;
        char* dst = smem + RB_P + (k & 1) * RB_RAWK * 1024;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int piece = w4 + 4 * i;
            if (piece < RB_RAWK) {
                const int row = 4 * piece + (lane >> 4);
                const int f = min(max(t0 - 2 + row, 0), p.T - 1);
                const bf16_t* src = p.xin + ((int64_t)b * p.T + f) * 128 + 8 * ((lane & 15) ^ RB_SWZ(row));
                __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(dst + piece * 1024), 16, 0, CPOL_NT);      // (read once: non-temporal)
            }
        }
        if (GATE && w4 == 3) {                                    // the utterance's gate row: 128 floats = 32 lanes x 16 B (the other lanes re-read it)
            const float* src = p.gate + (int64_t)b * 128 + 4 * (lane & 31);
            __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(smem + RB_GATE + (k & 1) * 1024), 16, 0, 0);
        }
    };
    if (grp == 0) {
        if (n_mine > 0) issue_dma(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    lds_barrier();

#ifdef SVHIP_GEMM_DEBUG
    unsigned long long tsum[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;       // tools/rb_bench: work / wait cycles of phase 1 and phase 2
#define RB_STAMP(i) if (p.dbg) { const unsigned long long t_ = __builtin_readcyclecounter(); tsum[i] += t_ - tprev; tprev = t_; }
    if (p.dbg) tprev = __builtin_readcyclecounter();
#else
#define RB_STAMP(i)
#endif
    // group B: running column sums of the pooled rows of utterance cs_b (lane = 8 channels x the rows of its row group)
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int cs_b = n_mine > 0 ? first / p.ntiles : 0;
    // one partial row per (utterance, workgroup segment, wave): lanes l, l+16, l+32, l+48 hold the same channels
    auto flush_colsum = [&](int b) {
        const int seg = (int)blockIdx.x - (b * p.ntiles) / p.per_wg;             // which of the workgroups sharing utterance b this is
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            cs[e] += __shfl_xor(cs[e], 16, 64);
            cs[e] += __shfl_xor(cs[e], 32, 64);
        }
        if (lane < 16 && seg >= 0 && seg < p.nseg) {
            float* dst = p.colsum + (((int64_t)b * p.nseg + seg) * 4 + w4) * 128 + 8 * (lane & 15);
            *reinterpret_cast<f32x4*>(dst) = f32x4{cs[0], cs[1], cs[2], cs[3]};
            *reinterpret_cast<f32x4*>(dst + 4) = f32x4{cs[4], cs[5], cs[6], cs[7]};
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) cs[e] = 0.0f;
    };
    // round r: matrix phase A: conv1(item r) | B: conv2(item r - 1); vector phase A: convert(item r + 1), request(item r + 2) | B: pool(item r - 1)
    for (int r = -1; r <= n_mine; ++r) {
        // ============================================== matrix phase ==============================================
        if (grp == 0) {
            if (r >= 0 && r < n_mine) {
                // h = lrelu(bn2(conv1(pre))), zero outside [0, T) (conv2's zero padding); H row q <-> frame t0 - 1 + q
                const int item = first + r;
                const int b = item / p.ntiles, t0 = (item - b * p.ntiles) * RB_TT;
                const int r16 = opaque(r16_), q4 = opaque(q4_);
                f32x4 sc2[2], sh2[2];                             // epilogue constants of the lane's channels n = w4*32 + cb*16 + 4*q4 + e
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    sc2[cb] = *reinterpret_cast<const f32x4*>(smem + RB_CST + (3 * 128 + w4 * 32 + cb * 16 + 4 * q4) * 4);
                    sh2[cb] = *reinterpret_cast<const f32x4*>(smem + RB_CST + (4 * 128 + w4 * 32 + cb * 16 + 4 * q4) * 4);
                }
                f32x4 acc[RB_FB][2];
                conv_k3<false, H>(smem + RB_P + (r & 1) * RB_RAWK * 1024, wf, acc, r16, q4, nullptr, w4, p.debug);
                RB_STAMP(4)
                char* hbuf = smem + RB_H + (r & 1) * RB_HR * 256;
                const bool edge = (t0 == 0) || (t0 + RB_TT + 2 > p.T);            // workgroup-uniform
#pragma unroll
                for (int fb = 0; fb < RB_FB; ++fb) {
                    const int q = fb * 16 + r16;
                    const int f = t0 - 1 + q;
                    const uint32_t vmask = (!edge || (f >= 0 && f < p.T)) ? 0xffffffffu : 0u;
                    u32x2 o[2];
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) {
                        o[cb][0] = bf_pack(lrelu03(fmaf(acc[fb][cb][0], sc2[cb][0], sh2[cb][0])), lrelu03(fmaf(acc[fb][cb][1], sc2[cb][1], sh2[cb][1]))) & vmask;
                        o[cb][1] = bf_pack(lrelu03(fmaf(acc[fb][cb][2], sc2[cb][2], sh2[cb][2])), lrelu03(fmaf(acc[fb][cb][3], sc2[cb][3], sh2[cb][3]))) & vmask;
                    }
                    // whole 16-byte chunks: the even lane of a pair writes channel block 0 (its half, then its partner's), the odd
                    // one block 1 (see conv_k3's accumulator load) — ds_write_b128 of 8 consecutive rows is conflict-free
                    const auto s0 = __builtin_amdgcn_permlane16_swap(o[0][0], o[1][0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(o[0][1], o[1][1], false, false);
                    const int c16 = w4 * 4 + (q4 & 1) * 2 + (q4 >> 1);
                    *reinterpret_cast<u32x4*>(hbuf + q * 256 + ((c16 ^ RB_SWZ(q)) << 4)) = u32x4{s0[0], s1[0], s0[1], s1[1]};
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the rows requested in the previous vector phase have landed (converted next)
        } else if (r >= 1) {
            // o = y + conv2(h) (identity shortcut: the accumulators start from y) -> O; row q <-> frame t0 + q
            const int k = r - 1;
            const int r16 = opaque(r16_), q4 = opaque(q4_);
            f32x4 acc[RB_FB][2];
            conv_k3<true, H>(smem + RB_H + (k & 1) * RB_HR * 256, wf, acc, r16, q4, smem + RB_Y + (k & 1) * RB_YR * 256, w4, p.debug);
            RB_STAMP(4)
#pragma unroll
            for (int fb = 0; fb < RB_FB; ++fb) {
                const int q = fb * 16 + r16;
                const auto s0 = __builtin_amdgcn_permlane16_swap(bf_pack(acc[fb][0][0], acc[fb][0][1]), bf_pack(acc[fb][1][0], acc[fb][1][1]), false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(bf_pack(acc[fb][0][2], acc[fb][0][3]), bf_pack(acc[fb][1][2], acc[fb][1][3]), false, false);
                const int c16 = w4 * 4 + (q4 & 1) * 2 + (q4 >> 1);
                *reinterpret_cast<u32x4*>(smem + RB_O + q * 256 + ((c16 ^ RB_SWZ(q)) << 4)) = u32x4{s0[0], s1[0], s0[1], s1[1]};
            }
        }
        RB_STAMP(0)
        lds_barrier();                                            // H[r & 1], O and the landed rows are visible; P[r & 1] is no longer read
        RB_STAMP(1)
        // ============================================== vector phase ==============================================
        // y = (xin + alpha) * gate (rounded to bf16, as a stored tensor would be), pre = lrelu(bn1(y)), in place: the thread owns
        // data chunk c16 of rows row0 + rg + 16 i, i < NI.  Branch-free: a per-element `valid ? f(x) : 0` compiles to eight
        // exec-mask branches with an LDS wait each.  Group A converts rows 0..63, group B (whose pooling is half as long) the
        // rest: with all 82 rows on group A its waves were the vector phase's critical path (3.2 k of 9.9 k cycles per round
        // while group B waited 1.7 k at the barrier).
        auto convert = [&](auto ni_tag, int row0) {
            constexpr int NI = decltype(ni_tag)::value;
            const int k = r + 1, item = first + k;
            const int b = item / p.ntiles, t0 = (item - b * p.ntiles) * RB_TT;
            (void)b;
            const int c16 = opaque(c16_), rg = opaque(rg_);
            f32x4 sc1[2], sh1[2], al[2], gt[2];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                sc1[hf] = *reinterpret_cast<const f32x4*>(smem + RB_CST + (0 * 128 + 8 * c16 + 4 * hf) * 4);
                sh1[hf] = *reinterpret_cast<const f32x4*>(smem + RB_CST + (1 * 128 + 8 * c16 + 4 * hf) * 4);
                if (GATE) {
                    al[hf] = *reinterpret_cast<const f32x4*>(smem + RB_CST + (2 * 128 + 8 * c16 + 4 * hf) * 4);
                    gt[hf] = *reinterpret_cast<const f32x4*>(smem + RB_GATE + (k & 1) * 1024 + (8 * c16 + 4 * hf) * 4);
                }
            }
            char* pbuf = smem + RB_P + (k & 1) * RB_RAWK * 1024;
            char* ybuf = smem + RB_Y + (k & 1) * RB_YR * 256;
            u32x4 xin[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int row = min(row0 + rg + 16 * i, RB_PR - 1);
                xin[i] = *reinterpret_cast<const u32x4*>(pbuf + row * 256 + ((c16 ^ RB_SWZ(row)) << 4));
            }
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int row = row0 + rg + 16 * i;
                const int f = t0 - 2 + row;
                const uint32_t vmask = (f >= 0 && f < p.T) ? 0xffffffffu : 0u;        // conv1's zero padding applies to `pre`
                u32x4 y4, pre4;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    float a = bf_lo(xin[i][w]), c = bf_hi(xin[i][w]);
                    uint32_t yw = xin[i][w];
                    if (GATE) {
                        yw = bf_pack((a + al[w >> 1][(2 * w) & 3]) * gt[w >> 1][(2 * w) & 3], (c + al[w >> 1][(2 * w + 1) & 3]) * gt[w >> 1][(2 * w + 1) & 3]);
                        a = bf_lo(yw); c = bf_hi(yw);
                    }
                    y4[w] = yw;
                    pre4[w] = bf_pack(lrelu03(fmaf(a, sc1[w >> 1][(2 * w) & 3], sh1[w >> 1][(2 * w) & 3])),
                                      lrelu03(fmaf(c, sc1[w >> 1][(2 * w + 1) & 3], sh1[w >> 1][(2 * w + 1) & 3]))) & vmask;
                }
                if (row < RB_PR) {
                    *reinterpret_cast<u32x4*>(pbuf + row * 256 + ((c16 ^ RB_SWZ(row)) << 4)) = pre4;
                    if (row >= 2) {
                        const int yr = row - 2;
                        *reinterpret_cast<u32x4*>(ybuf + yr * 256 + ((c16 ^ RB_SWZ(yr)) << 4)) = y4;
                    }
                }
            }
        };
        const bool conv_next = r + 1 < n_mine && !(p.debug & 4);
        if (grp == 0) {
            if (conv_next) convert(RbInt<4>{}, 0);
            if (r + 2 < n_mine) issue_dma(r + 2);                 // into P[r & 1], which conv1(r) has finished with; converted next round
        } else if (r >= 1 && !(p.debug & 8)) {
            // max_pool1d(3) of O -> global, and this wave's column sums of the pooled rows
            const int item = first + r - 1;
            const int b = item / p.ntiles, tile = item - b * p.ntiles;
            const int t0 = tile * RB_TT;
            const int c16 = opaque(c16_), rg = opaque(rg_);
            if (b != cs_b) { flush_colsum(cs_b); cs_b = b; }
            u32x4 a[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int j = min(rg + 16 * i, RB_TT / 3 - 1);    // pooled row of the tile (clamped: the store below is what is guarded)
#pragma unroll
                for (int s3 = 0; s3 < 3; ++s3) {
                    const int q = 3 * j + s3;
                    a[i][s3] = *reinterpret_cast<const u32x4*>(smem + RB_O + q * 256 + ((c16 ^ RB_SWZ(q)) << 4));
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int j = rg + 16 * i;
                const int tp = t0 / 3 + j;
                const bool ok = j < RB_TT / 3 && tp < p.Tout;
                const float keep = ok ? 1.0f : 0.0f;
                u32x4 m;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const float lo = fmaxf(fmaxf(bf_lo(a[i][0][w]), bf_lo(a[i][1][w])), bf_lo(a[i][2][w]));
                    const float hi = fmaxf(fmaxf(bf_hi(a[i][0][w]), bf_hi(a[i][1][w])), bf_hi(a[i][2][w]));
                    m[w] = bf_pack(lo, hi);                       // the max of bf16 values is a bf16 value: exact
                    cs[2 * w] = fmaf(lo, keep, cs[2 * w]);
                    cs[2 * w + 1] = fmaf(hi, keep, cs[2 * w + 1]);
                }
                if (ok) *reinterpret_cast<u32x4*>(p.opool + ((int64_t)b * p.Tout + tp) * 128 + 8 * c16) = m;
            }
        }
        if (grp == 1 && conv_next) convert(RbInt<2>{}, 64);       // rows 64 .. RB_PR - 1 of the next item
        RB_STAMP(2)
        lds_barrier();                                            // P[(r + 1) & 1] (converted) and Y are complete; O is no longer read
        RB_STAMP(3)
    }
    if (grp == 1 && n_mine > 0) flush_colsum(cs_b);
#ifdef SVHIP_GEMM_DEBUG
    if (p.dbg && lane == 0 && w4 == 0)
        for (int i = 0; i < 6; ++i) p.dbg[((int64_t)blockIdx.x * 2 + grp) * 6 + i] = tsum[i];
#endif
#undef RB_STAMP
}
#undef bf_lo
#undef bf_hi
#undef bf_pack

// AFMS gate: s[b, n] = sigmoid(bias[n] + sum_c W[n, c] * mean[b, c]), mean = (sum of the partial column sums) / Tn
// (RawNet_baseline.py:64-66): a (B x C) x (C x C) product too small for the GEMM kernels and, done one utterance per workgroup,
// bound by re-reading the weight matrix per utterance (40 us).  Workgroup = 16 outputs x 16 utterances: both operand slices are
// staged in LDS with every load in flight at once (WT = the fc weight transposed, so a slice is 64-byte row pieces), then
// thread = (output, utterance) runs the dot product out of LDS.
// (GATE_U = 4 for small batches: four times the workgroups, and a thread sums the partial rows of 2 (c, utterance) pairs instead of 8 —
//  at B = 20 the fused blocks hand over 56 partial rows per utterance and the kernel took 29 us)
constexpr int GATE_N = 16;
template <int GATE_U>
__global__ __launch_bounds__(256) void rn_afms_gate_kernel(const float* __restrict__ part, int nparts, int B, int C, float inv_T,
                                                           const float* __restrict__ WT, const float* __restrict__ bias,
                                                           float* __restrict__ s) {
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    float* wl = gsm;                       // [C][GATE_N]
    float* mt = gsm + C * GATE_N;          // [C][GATE_U]
    const int n0 = blockIdx.x * GATE_N, b0 = blockIdx.y * GATE_U, tid = threadIdx.x;
    // weight slice: 4 lanes x 16 bytes per c row, 64 rows per pass
    {
        const int q = tid & 3, r = tid >> 2;
        f32x4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = r + 64 * i;
            v[i] = c < C ? *reinterpret_cast<const f32x4*>(WT + (int64_t)c * C + n0 + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = r + 64 * i;
            if (c < C) *reinterpret_cast<f32x4*>(wl + c * GATE_N + 4 * q) = v[i];
        }
    }
    // mean slice, transposed to [c][u]: consecutive threads read consecutive c of one partial row
    for (int i = tid; i < C * GATE_U; i += 256) {
        const int u = i / C, c = i - u * C;
        float a = 0.0f;
        if (b0 + u < B) {
            const float* src = part + (int64_t)(b0 + u) * nparts * C + c;
#pragma unroll 8
            for (int t = 0; t < nparts; ++t) a += src[(int64_t)t * C];
        }
        mt[c * GATE_U + u] = a * inv_T;
    }
    __syncthreads();
    const int n = tid & 15, u = tid >> 4;
    if (u >= GATE_U) return;
    float acc = 0.0f;
#pragma unroll 8
    for (int c = 0; c < C; ++c) acc = fmaf(wl[c * GATE_N + n], mt[c * GATE_U + u], acc);
    if (b0 + u < B) s[(int64_t)(b0 + u) * C + n0 + n] = 1.0f / (1.0f + expf(-(acc + bias[n0 + n])));
}

// The same gate for full batches of wide blocks (B > 64, C = 256 / 512: the late block tails) on the exact fp32 MFMA: workgroup = 32 outputs x 32
// utterances, its four waves split the C input channels, a lane's 16-byte load of the mean (summed from the partial rows on the fly) supplies
// four MFMA steps and the matching weight rows are read with one coalesced 4-byte load each.  (The LDS-staged VALU kernel above: 28 us at
// C = 512, B = 256; the one-workgroup-per-utterance tail streams the whole 1 MiB matrix through one CU per utterance: ~25 us.)
__global__ __launch_bounds__(256) void rn_afms_gate_mfma_kernel(const float* __restrict__ part, int nparts, int B, int C, float inv_T,
                                                                const float* __restrict__ WT, const float* __restrict__ bias,
                                                                float* __restrict__ s) {
    __shared__ float red[3][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
    const int kper = C >> 2;                               // input channels per wave
    const int kb = wave * kper + 4 * h;
    const int b = min(b0 + r, B - 1);
    const float* prow = part + (int64_t)b * nparts * C + kb;
    const float* wcol = WT + (int64_t)kb * C + n0 + r;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    for (int j = 0; j < kper / 8; ++j) {                  // groups of 8 input channels: this lane's four are 8 j + 4 h + e
        f32x4 m = *reinterpret_cast<const f32x4*>(prow + 8 * j);
        for (int t = 1; t < nparts; ++t) m += *reinterpret_cast<const f32x4*>(prow + (int64_t)t * C + 8 * j);
        m *= inv_T;
        float w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = wcol[(int64_t)(8 * j + e) * C];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[e], m[e], acc, 0, 0, 0);
    }
    if (wave > 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) red[wave - 1][e][lane] = acc[e];
    }
    __syncthreads();
    if (wave == 0 && b0 + r < B) {
        // acc[e]: output n0 + (e & 3) + 8 (e >> 2) + 4 h of utterance b0 + r
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = n0 + 8 * g + 4 * h;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + n);
            f32x4 v;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float a = ((acc[4 * g + u] + red[0][4 * g + u][lane]) + red[1][4 * g + u][lane]) + red[2][4 * g + u][lane] + bv[u];
                v[u] = 1.0f / (1.0f + expf(-a));
            }
            *reinterpret_cast<f32x4*>(s + (int64_t)(b0 + r) * C + n) = v;
        }
    }
}

}  // namespace

int rn_block128_ntiles(int T) { return (3 * (T / 3) + RB_TT - 1) / RB_TT; }
// items per workgroup and the number of workgroups that can share one utterance (partial column-sum rows per utterance = 4 * nseg)
static void rn_block128_split(int B, int T, int num_cu, int* per_wg, int* nseg, int* grid) {
    const int nt = rn_block128_ntiles(T), items = B * nt;
    const int g = items < num_cu ? items : num_cu;
    const int per = (items + g - 1) / g;
    *per_wg = per;
    *grid = (items + per - 1) / per;
    *nseg = (nt + per - 1) / per + 1;
}
int rn_block128_nparts(int B, int T, int num_cu) {
    int per, nseg, grid;
    rn_block128_split(B, T, num_cu, &per, &nseg, &grid);
    return 4 * nseg;
}

bool rn_block128_supported(int cin, int cout, int T, bool downsample, bool has_shortcut, int Kp1, int Kp2) {
    return cin == 128 && cout == 128 && downsample && !has_shortcut && Kp1 == 384 && Kp2 == 384 && T >= 3;
}

hipError_t launch_rn_block128(const RnBlock128Params& p_in, int num_cu, hipStream_t stream) {
    RnBlock128Params p = p_in;
    if (!p.xin || !p.W1 || !p.W2 || !p.opool || !p.colsum || p.B <= 0 || p.T < 3 || p.Tout != p.T / 3 || p.ntiles != rn_block128_ntiles(p.T))
        return hipErrorInvalidValue;
    if ((p.alpha == nullptr) != (p.gate == nullptr)) return hipErrorInvalidValue;
    int grid = 0;
    rn_block128_split(p.B, p.T, num_cu, &p.per_wg, &p.nseg, &grid);
    // the partial rows of segments an utterance does not have stay zero
    if (hipError_t e = hipMemsetAsync(p.colsum, 0, (size_t)p.B * p.nseg * 4 * 128 * sizeof(float), stream)) return e;
#define SV_RB(G, HH)                                                                                                       \
    {                                                                                                                      \
        static DeviceOnce attr;                                                                                            \
        if (hipError_t e = set_max_dynamic_lds(attr, reinterpret_cast<const void*>(rn_block128_kernel<G, HH>), RB_LDS)) return e; \
        hipLaunchKernelGGL((rn_block128_kernel<G, HH>), dim3(grid), dim3(512), RB_LDS, stream, p);                         \
    }
    if (p.f16) { if (p.gate) SV_RB(true, f16_t) else SV_RB(false, f16_t) }
    else { if (p.gate) SV_RB(true, bf16_t) else SV_RB(false, bf16_t) }
#undef SV_RB
    return hipGetLastError();
}

hipError_t launch_rn_afms_gate(const float* part, int nparts, int B, int C, int Tn, const float* WT, const float* bias, float* s,
                               hipStream_t stream) {
    if (!part || !WT || !bias || !s || C > 512 || C % 64 != 0 || nparts <= 0 || Tn <= 0 || B <= 0) return hipErrorInvalidValue;
    if (B > 64 && C >= 256 && C % 32 == 0 && nparts <= 16 && ((reinterpret_cast<uintptr_t>(part) | reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(s)) & 15) == 0) {
        hipLaunchKernelGGL(rn_afms_gate_mfma_kernel, dim3(C / 32, (B + 31) / 32), dim3(256), 0, stream, part, nparts, B, C, 1.0f / (float)Tn, WT, bias, s);
        return hipGetLastError();
    }
    if (B <= 64) {
        const size_t lds = (size_t)C * (GATE_N + 4) * sizeof(float);
        hipLaunchKernelGGL(rn_afms_gate_kernel<4>, dim3(C / GATE_N, (B + 3) / 4), dim3(256), lds, stream, part, nparts, B, C, 1.0f / (float)Tn, WT, bias, s);
    } else {
        const size_t lds = (size_t)C * (GATE_N + 16) * sizeof(float);       // <= 64 KiB
        hipLaunchKernelGGL(rn_afms_gate_kernel<16>, dim3(C / GATE_N, (B + 15) / 16), dim3(256), lds, stream, part, nparts, B, C, 1.0f / (float)Tn, WT, bias, s);
    }
    return hipGetLastError();
}

}  // namespace svhip
