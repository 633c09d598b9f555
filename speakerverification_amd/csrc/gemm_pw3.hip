// gemm_pw3.hip — PERSISTENT bf16 pointwise GEMM, 256 x 256 tile, epilogue straight from the accumulators (gfx950).
//
// Same contract and the same four-phase K loop as gemm_pw2.hip (Y = epi(A . W^T), bf16 in / fp32 accumulate / bf16 out; read that
// file's header for the ring, the swizzle and the one-phase stagger of the two wave groups).  What differs is everything
// around the K loop, which on the K = 1024 layers of ECAPA-TDNN was a third of a tile's time (DMA prologue 6 %, GELU epilogue +
// LDS staging 20 %, stores 7 %, gap to the next workgroup 3 %):
//
//   * one workgroup per CU walks a list of tiles (round r: tile r * G + band position, so that the 32 CUs of an XCD work on
//     8 M-tiles x 4 N-tiles at a time and share their X and W lines in that XCD's L2);
//   * the epilogue never touches the LDS ring: activation / BN affine on the accumulators, bf16 pairs, one v_permlane16_swap per
//     dword so that a lane owns 16 contiguous bytes, and 16-byte global stores (16 frames x 64 bytes per wave instruction);
//     the per-utterance column sums (SE squeeze, ASP statistics) are DPP row reductions of the same registers (fp32, unrounded);
//   * so the DMA stream never stops: in a tile's last two K tiles the phases whose issue slot would be empty fetch the NEXT
//     tile's first seven half-tiles (K tile count even: the buffer parities simply continue), with the stream's ordinary
//     vmcnt(10) waits; its 3 KiB of bias / scale / shift go to a double-buffered LDS strip by DMA as well, so no ordinary
//     global load sits beside the DMAs.  The epilogue runs with five half-tiles in flight and the tile's stores drain under
//     the next tile's K loop;
//   * vmcnt bookkeeping: a wave's queue at the next tile's start is [14 operand DMAs][NST stores], in that order (pinned with
//     sched_barrier), so the first five phases of a tile wait with vmcnt(10 + NST) instead of vmcnt(10) — the counter retires
//     in order, and a strict count would wait for the store acknowledgements; after a tile with masked rows (a wave may have
//     issued fewer stores) and on a workgroup's first tile the strict count is used.
//
// What it bought (round 3, tools/gemm_bench + stamps, profiles/r03_pw3_*): the prologue, the stores and the inter-workgroup
// gap are gone from the tile (start wait 1.2 k of 53.5 k cycles), the K loop is unchanged (40.5 k cycles for 32.8 k of MFMA
// issue), and what remains between two K loops is the GELU itself: 12 - 13 k cycles in which the two waves of a SIMD share one
// vector pipe (v_pk_* and v_exp / v_rcp cost 8 issue cycles per wave instruction, the rest 4: ~47 cycles per 64 outputs per
// wave, and the two waves' streams add, they do not overlap).  K = 1024 layers 0.245 - 0.26 ms (per-tile kernel 0.255 - 0.27;
// with the column sums 0.25 against 0.30), K = 3072 1.50 against 1.57 ms.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "kernels.h"
#include "gemm_epi.h"

namespace svhip {

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

constexpr int HT = 16384;                       // one half-tile: 128 rows x 64 k bf16
constexpr int RING = 8 * HT;                    // 128 KiB operand ring
constexpr int CST = 3072;                       // per-tile constants: bias | scale | shift, 256 floats each
constexpr int PW3_LDS = RING + 2 * CST;
constexpr int PGROUP_M = 8;                     // M-tiles per tile group: 8 x 4 N-tiles = the 32 CUs of an XCD

#ifdef SVHIP_GEMM_DEBUG
constexpr bool DBG3 = true;      // tools/gemm_bench (debug bit 16384): per-workgroup stage cycle totals
#else
constexpr bool DBG3 = false;
#endif
// compile-time ablations (tools/abl_pw3.sh; 0 in every shipped build): 1 no MFMAs, 2 no operand DMAs, 4 no activation,
// 8 no output stores (without column sums the whole K loop is then dead code), 16 no fragment reads, 64 no column sums,
// 128 strict vmcnt everywhere (the stores' acknowledgements are waited for in the next tile's first phases)
#ifdef PW3_ABL
constexpr int ABL = PW3_ABL;
#else
constexpr int ABL = 0;
#endif

// X3 (SVHIP_F32X3 handles: fp32 storage, products as split-bf16 MFMA triples, ~2^-17 per product): A and W arrive in the "S32"
// layout — per row, per block of 32 k: 32 hi bf16 | 32 lo bf16 (hi = bf16(v), lo = bf16(v - hi); 128 bytes, the bytes of the fp32
// values they replace).  A 128-byte LDS row is then ONE 32-wide k block instead of a 64-wide one, the DMA stream, the ring, the
// swizzle and the fragment reads are those of the bf16 kernel unchanged ("ks 0" reads the hi fragment, "ks 1" the lo one), and a
// fragment pair takes three MFMAs (hi.hi + hi.lo + lo.hi) instead of two: 1.5 x the matrix work per byte staged, no conversion
// instruction anywhere in the loop.  The epilogue is fp32: exact erf GELU, 16-byte fp32 stores straight from the accumulators.
//
// R2 (X3 only): one step of a Res2Net chain (models/ECAPA_TDNN.py:118-129) — y_j = BN(ReLU(conv_k3_dilated(U_j))) with U_j =
// c_j + y_{j-1} already in the S32 layout (M, cin): the X half-tiles are the im2col view of the dilated convolution (a 128-byte K
// tile lies inside ONE tap, so the tap shift is uniform per K tile and only the per-lane source row moves: reflect(t + (tap - 1)
// dil) inside the lane's utterance); N = cin <= 128 fills the two left wave columns of the 256 x 256 tile (the right two
// multiply clamped weight rows and store nothing); the epilogue writes y_j in the S32 layout into the chain output (the next
// GEMM's A operand: no fp32 copy exists) and U_{j+1} = y_j + c_{j+1} (c from the fp32 tdnn1 output) into the next step's input.
// Measured (round 3, C = 1024, B = 256; tools/r2_bench with the kernel's stage stamps): 88 us per step, 115 TFLOP/s of reference
// FLOPs.  Per tile: K loop 44 k cycles (12 K tiles; the matrix pipe is 83 % busy in it, half of that on the empty wave columns),
// epilogue 37 k, start 4.5 k; 401 tiles are 1.57 per workgroup, i.e. two rounds.  History of the epilogue: c_next loaded straight
// into registers between the stores 52 k cycles (every load's wait also waited for all earlier stores); an L2 prefetch of the c
// rows by LDS-DMA no gain; the c tile through the ring + 16-byte swapped stores from the accumulator layout 35.6 k, bound by the
// CU's store path (64 separate 16-byte pieces per wave instruction), 100 us per step; U and y staged through the ring as whole
// rows and copied out by all eight waves (1 KiB contiguous per instruction): the same cycle count at a higher clock, 88 us.
// What bounds it now: every CU reaches its epilogue at the same time, and 384 KB per tile (c in, y and U out) at a CU's share of
// the HBM bandwidth (~10 B / cycle) IS ~38 k cycles, while HBM idles through the K loops; overlapping the two needs two tiles
// in flight per CU (a 128 x 128 / four-wave re-cut with two workgroups per CU), not built.  Letting the two empty wave columns skip
// their fragment reads and MFMAs (they keep the DMA stream, waits and barriers) changed the step by nothing (88.7 us): the clock rose
// and the K loop's cycle count with it (44 k -> 52 k) — that loop runs at the pace of its gathered operand stream, not of the matrix pipe.
//
// CV on 16-bit operands (round 4; H = bf16_t or f16_t): the same gather with 64-wide K tiles — a 128-byte K tile is 64 channels of ONE
// tap, so cin % 64 == 0 (RawNet2's k = 3 convolutions of blocks 2 - 5: cin = 128 / 256 / 512; ECAPA's blocks.0 with its 80 mel
// channels zero-padded to 128).  Zero padding (RawNet2) as well as reflect: a row outside its utterance reads the zero page (a
// 32-bit offset from the K tile's base: the host places the zero page behind the operand).  No residual operand: an identity shortcut is
// added by the consumer (rn_tail), so that this kernel's epilogue stays free of ordinary loads.  Measured at B = 256 (fp16, in the model /
// tools/gemm_bench back to back): block 2 conv1 (K = 384) 90 us against 100 on the per-tile kernel (719 against 622 TFLOP/s back to
// back), the K = 768 convolutions of blocks 3 / 4 62.6 against 69.6 us.  What bounds these shapes is not the gather (the plain pointwise
// GEMM of the same shape runs at 784 TFLOP/s): N = 256 is one N tile, so no operand is reused between tiles, and 391 tiles on 256 CUs
// are 1.53 rounds.  A form with the 1 x 1 shortcut of blocks 2 / 5 appended to conv2's K axis (as gemm_pw2's CONV == 2) was built and
// measured no faster than that kernel (193.6 against 182.0 us, 129.6 against 119.4): those two launches stay on gemm_pw2.
template <int EPI, int CS, bool X3, bool R2 = false, bool CV = false, typename H = bf16_t>          // CS: 0 no column sums, 1 sums, 2 sums and sums of squares
__global__ __launch_bounds__(512, 2) void gemm_pw3_kernel(GemmParams p) {
    static_assert(!R2 || (X3 && CS == 0), "the Res2Net step form exists for the X3 kernel only");
    static_assert(!CV || !R2, "the conv-gather form (CV) is the pointwise kernel with gathered X rows");
    typedef typename std::conditional<X3, x3_t, H>::type MH;       // MFMA operand type: the split forms' planes are x3_t (common.h)
    constexpr bool GATHER = R2 || CV;          // X half-tiles = im2col view of a dilated convolution over the rows of an utterance
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NST = (X3 ? 32 : 16) + 8 * CS;    // vector-memory stores a wave issues in one tile's epilogue
    constexpr int ESZ = X3 ? 4 : 2;                 // bytes per k of an operand row

    const int ntm = (p.M + 255) / 256, ntn = (p.N + 255) / 256;
    const int ntiles = ntm * ntn;
    const int G = gridDim.x;
    int perm = blockIdx.x;                          // position in a round: contiguous band per XCD (bijective for any G)
    {
        const int q = G >> 3, r = G & 7, xcd = perm & 7;
        perm = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (perm >> 3);
    }
    // Tail split (round 5; 16-bit pointwise / conv-gather forms): ntiles = q G + R with 0 < R <= G / 2 leaves the last round with R busy
    // workgroups and G - R idle ones for a whole tile time (the K = 1024 layers of ECAPA at B = 256: 1 604 tiles on 256 CUs = 6.27
    // rounds, run as 7).  The R tail tiles are then walked as 2 R HALF items, one per workgroup: the tile's 256 frames x 128 of its 256
    // channels — the W-lo column set of the four-phase K tile (phases 0 and 3: X-lo x W-lo, X-hi x W-lo) for the first half, and the
    // same code on a channel origin 32 higher for the second (its "W-lo" rows are then the tile's W-hi set).  Phases 1 and 2 keep their
    // DMA issue, waits, barriers and fragment reads (a skipped W-hi read would keep the old fragments alive across the loop: 16 VGPRs, and
    // the column-sum instances spilled) and skip their MFMAs; the epilogue runs for j < 2.  Every output element and every
    // column sum sees the same products in the same order as in a whole tile: bit-identical results.
    constexpr bool SPLIT_OK = !R2 && !(X3 && CV);     // (the X3 conv-gather form keeps whole tiles: its y_s32 output is addressed per tile)
    const int qfull = (ntiles / G) * G;
    const int rtail = ntiles - qfull;
    // (qfull == 0: a grid of 2 * ntiles workgroups over ntiles <= CUs / 2 tiles — a small batch — walks nothing but halves, one each)
    const bool split = SPLIT_OK && p.tail_split && rtail > 0 && 2 * rtail <= G;
    const int nitems = split ? qfull + 2 * rtail : ntiles;
    // item w -> tile index and half selector (0: whole tile, 1: first column half, 2: second)
    auto item_tile = [&](int w, int& hsel) {
        if (!split || w < qfull) { hsel = 0; return w; }
        hsel = 1 + ((w - qfull) & 1);
        return qfull + ((w - qfull) >> 1);
    };
    auto tile_of = [&](int w, int& tm, int& tn) {
        const int per = PGROUP_M * ntn;
        const int grp = w / per;
        const int within = w - grp * per;
        const int gm = min(PGROUP_M, ntm - grp * PGROUP_M);
        const int tnn = within / gm;
        tm = grp * PGROUP_M + (within - tnn * gm);
        tn = tnn;
    };

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;        // wave group == wm: rows wm*128 .. +127, cols wn*64 .. +63
    // Only `tid` stays live across the stages of a tile: every stage derives its lane coordinates from an OPAQUE copy, so that
    // the compiler cannot hoist a stage's address arithmetic out of the tile loop and carry it through the K loop (which runs
    // at 256 VGPRs: every hoisted value is a spill, and every spill reload is a vector-memory load that drains the DMA queue)
    auto lane_now = [&]() { int t = tid; asm volatile("" : "+v"(t)); return t & 63; };

    // ---- operand DMA addressing (see gemm_pw2.hip) ----------------------------------------------------------------
    // a lane's source = wave-uniform tile base (SGPR pair: A + m0 rows, W + n0 rows, advanced by 128 bytes per K tile) + a 32-bit
    // per-lane byte offset (row within the tile, clamped to the matrix, and the swizzled 16-byte chunk): 8 VGPRs, no 64-bit
    // vector adds in the loop
    uint32_t xo[2][2], wo[2][2];
    uint32_t xt[2] = {0, 0};                       // R2: frame index (inside its utterance) of the lane's X rows, [jj] = lo | hi << 16
    const char* abase = nullptr;
    const char* wbase = nullptr;
    const bool zero_pad = CV && !X3 && p.pad_mode == PAD_ZERO;
    int dsto[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) dsto[jj] = (wave * 2 + jj) * 1024;
    auto set_src = [&](int m0, int n0) {
        const int lane = lane_now();
        // (R2: the gathered rows lie up to 2 dil <= 8 rows before the lane's own; the base sits 8 rows low so that offsets stay >= 0)
        abase = reinterpret_cast<const char*>(p.A) + ((int64_t)m0 - (GATHER ? 8 : 0)) * p.lda * ESZ;
        wbase = reinterpret_cast<const char*>(p.W) + (int64_t)n0 * p.Kp * ESZ;
        const int mmax = p.M - 1 - m0, nmax = p.Wrows - 1 - n0;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int rho = (wave * 2 + jj) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((rho >> 1) & 7);
#pragma unroll
            for (int ty = 0; ty < 2; ++ty) {
                const int m = min((rho >> 6) * 128 + ty * 64 + (rho & 63), mmax);
                // (24-bit multiplies — v_mad_u32_u24, a 32-bit result: as a plain 32-bit product hipcc emits v_mad_u64_u32 and keeps the
                //  dead high half allocated with the low one, a second VGPR per offset; rows and row bytes < 2^24: pw3_offsets_fit)
                xo[ty][jj] = __umul24((uint32_t)(m + (GATHER ? 8 : 0)), (uint32_t)(p.lda * ESZ)) + (uint32_t)(c * 16);
                if (GATHER) {
                    // frame inside its utterance = (m0 + m) mod T by the host's multiplier ceil(2^32 / T) (GemmParams::t_magic, an SGPR):
                    // the quotient is exact or one too large, fixed by one add.  (Written as `% p.T` the compiler keeps a per-lane
                    // reciprocal of T alive across the whole tile loop — one VGPR too many for the 16-bit conv-gather instances,
                    // whose spill reload sat inside the K loop.)
                    const uint32_t mm = (uint32_t)(m0 + m);
                    int t = (int)(mm - __umulhi(mm, p.t_magic) * (uint32_t)p.T);
                    t = t < 0 ? t + p.T : t;
                    xt[jj] = ty == 0 ? (uint32_t)t : (xt[jj] | ((uint32_t)t << 16));
                }
                const int n = min((rho >> 5) * 64 + ty * 32 + (rho & 31), nmax);
                wo[ty][jj] = __umul24((uint32_t)n, (uint32_t)(p.Kp * ESZ)) + (uint32_t)(c * 16);
            }
        }
    };
    // K tiles per tap = cin / 32 (X3: 32 k per 128-byte K tile) or cin / 64 (16-bit operands): p.cin is the channel count of a row of
    // A; tap = kt / ktpt by multiplication (kt < 64); K tiles past the last tap (padding of the K tile count to an even number)
    // re-read the last tap: their weights are 0
    const int ktpt = GATHER ? (p.cin * ESZ) >> 7 : 1;
    const int tap_mul = GATHER ? 65536 / ktpt + 1 : 0;
    auto issue = [&](int ty, int kt) {
        char* base = smem + ((kt & 1) * 4 + ty) * HT;
        const bool gather = GATHER && ty < 2;
        const int tap = gather ? min((kt * tap_mul) >> 16, p.taps - 1) : 0;
        const char* ub = (ty < 2 ? abase : wbase) + (int64_t)(gather ? min(kt - tap * ktpt, ktpt - 1) : kt) * 128;
        const int shift = gather ? (tap - (p.taps >> 1)) * p.dil : 0;
        // zero padding (16-bit CV form): a row outside its utterance reads the zero page, addressed as a 32-bit offset from this K tile's
        // base like every other row (the host puts the zero page behind A, within 4 GiB: gemm_pw3cv16_supported), so the select is one
        // v_cndmask on the offset and the SGPR base + VGPR offset addressing stays
        const uint32_t zoff = (CV && !X3) ? (uint32_t)(reinterpret_cast<uintptr_t>(p.zero_page) - reinterpret_cast<uintptr_t>(ub)) : 0u;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            uint32_t o = ty < 2 ? xo[ty][jj] : wo[ty - 2][jj];
            if (gather) {       // the row of frame reflect(t + shift) of the same utterance: a few rows either way (int32 arithmetic)
                const int t = (int)((xt[jj] >> (ty * 16)) & 0xffffu);
                if (CV && !X3) {
                    const int tt = t + shift;
                    const int d = zero_pad ? shift : reflect_idx(tt, p.T) - t;
                    const uint32_t oo = (uint32_t)((int)o + d * p.lda * ESZ);
                    o = (zero_pad && (unsigned)tt >= (unsigned)p.T) ? zoff : oo;
                } else {
                    const int d = reflect_idx(t + shift, p.T) - t;
                    o = (uint32_t)((int)o + d * p.lda * ESZ);
                }
            }
            asm volatile("" : "+v"(o));          // the zero-extension stays in this block: SGPR base + 32-bit VGPR offset addressing
            const char* s = ub + o;
            if (!(ABL & 2)) __builtin_amdgcn_global_load_lds((gbl_void*)s, (lds_void*)(base + dsto[jj]), 16, 0, 0);
        }
    };
    // bias / scale / shift of an N-tile -> constant strip `par`: one 16-byte-per-lane DMA each by waves 0, 1, 2 (older than
    // every operand DMA of the tile, so the tile's first counted wait covers them)
    auto issue_consts = [&](int n0, int par) {
        if (wave < 3) {
            const float* s = (wave == 0 ? p.bias : wave == 1 ? p.scale : p.shift) + min(n0 + lane_now() * 4, p.N - 4);      // (N < 256: the R2 form)
            __builtin_amdgcn_global_load_lds((gbl_void*)s, (lds_void*)(smem + RING + par * CST + wave * 1024), 16, 0, 0);
        }
    };
    auto issue_prologue = [&]() {
        issue(2, 0); issue(0, 0); issue(3, 0); issue(1, 0);
        issue(2, 1); issue(0, 1); issue(3, 1);
    };
    auto wait_left = [&](int left) {              // allow `left` half-tiles (2 DMAs each) of this wave to stay in flight
        if (left >= 5) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (left == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (left == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (left == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (left == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    // five half-tiles AND the previous tile's NST stores (issued after this tile's first 14 DMAs) may stay in flight
    auto wait_relaxed = [&]() {
        static_assert(NST == 16 || NST == 24 || NST == 32 || NST == 40 || NST == 48, "vmcnt(10 + NST)");
        if (NST == 16) asm volatile("s_waitcnt vmcnt(26)" ::: "memory");
        else if (NST == 24) asm volatile("s_waitcnt vmcnt(34)" ::: "memory");
        else if (NST == 32) asm volatile("s_waitcnt vmcnt(42)" ::: "memory");
        else if (NST == 40) asm volatile("s_waitcnt vmcnt(50)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(58)" ::: "memory");
    };

    const int nkt = p.Kp * ESZ / 128;             // 128-byte K tiles per row: >= 4 and even (host check)
    // X3 conv-gather form (the first convolution of an F32X3 handle): A holds s * x, s an exact power of two chosen on the device from the
    // input's max |x| (GemmParams::in_scale); the convolution is linear, so the accumulators start at s * bias and are multiplied back
    float s_in = 1.0f, inv_in = 1.0f;
    if (X3 && CV && p.in_scale) { s_in = p.in_scale[2]; inv_in = p.in_scale[1]; }

    unsigned long long tacc[4] = {0, 0, 0, 0}, tprev = 0;      // debug builds: cycles in {tile-start wait, K loop, next-tile issue, epilogue}
#define PW3_STAMP(i) if (DBG3 && (p.debug & 16384) && p.ts) { const unsigned long long t_ = __builtin_readcyclecounter(); tacc[i] += t_ - tprev; tprev = t_; }
    if (DBG3 && (p.debug & 16384) && p.ts) tprev = __builtin_readcyclecounter();
    const unsigned long long tstart = tprev;

    int w = perm;                                 // (the host launches G <= ntiles workgroups)
    int tm, tn, hsel;
    tile_of(item_tile(w, hsel), tm, tn);
    set_src(tm * 256, tn * 256 + (hsel == 2 ? 32 : 0));
    issue_consts(tn * 256 + (hsel == 2 ? 32 : 0), 0);
    issue_prologue();
    bool relaxed = false;                         // the queue holds exactly NST stores behind this tile's first 14 DMAs
    int ntile_done = 0;
    int kbias = 0;                                // K tiles >= nkt of the DMA stream belong to the next tile: its K tile index = k - kbias
    bf16x8 wlo[2][2];                             // W-lo fragments of the K tile about to be multiplied (carried from tile to tile)
    // ---- first tile only: W-lo(0), X-lo(0) (and the constants) of every wave have landed; later tiles' first half-tiles arrive
    // inside the previous tile's last two K tiles, with the stream's ordinary counted waits
    wait_left(5);
    __builtin_amdgcn_s_barrier();
    {
        const int lane = lane_now();
        const int r16 = lane & 15, q4 = lane >> 4;
        const int woff = (wn * 32 + r16) * 128, wkey = ((wn * 32 + r16) >> 1) & 7;
        const char* b = smem + 2 * HT;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) wlo[j][ks] = *reinterpret_cast<const bf16x8*>(b + woff + j * 2048 + (((ks * 4 + q4) ^ wkey) << 4));
    }

    for (int it = 0;; ++it) {
        const bool half = SPLIT_OK && hsel != 0;      // wave-uniform: this item is a column half of its tile
        const int m0 = tm * 256, n0 = tn * 256 + (hsel == 2 ? 32 : 0), par = it & 1;
        const char* cb = smem + RING + par * CST;
        const int w_next = w + G;
        const bool more = w_next < nitems;
        const bool pf = more && !R2;            // the stream runs on into the next tile (R2: its epilogue needs the ring, see there)
        int tm_n = 0, tn_n = 0, hsel_n = 0;
        if (more) tile_of(item_tile(w_next, hsel_n), tm_n, tn_n);
        const int n0_n = tn_n * 256 + (hsel_n == 2 ? 32 : 0);
        PW3_STAMP(0)
        if (wm == 1) __builtin_amdgcn_s_barrier();    // group 1 runs one phase behind group 0
        const int lane = lane_now();
        const int r16 = lane & 15, q4 = lane >> 4;      // 16x16x32 fragment coordinates
        const int xoff = (wm * 64 + r16) * 128, woff = (wn * 32 + r16) * 128;
        const int xkey = ((wm * 64 + r16) >> 1) & 7, wkey = ((wn * 32 + r16) >> 1) & 7;

        // accumulators acc16[i][j][e] = channel n0 + wn*64 + j*16 + 4*q4 + e of frame m0 + wm*128 + i*16 + r16; they start at the bias
        f32x4 acc16[8][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 b0 = *reinterpret_cast<const f32x4*>(cb + (wn * 64 + j * 16 + 4 * q4) * 4);
            if (X3 && CV) b0 *= s_in;
#pragma unroll
            for (int i = 0; i < 8; ++i) acc16[i][j] = b0;
        }
        bf16x8 xf[4][2], whi[2][2], wnx[2][2];
        // ---- four-phase K tiles (gemm_pw2.hip).  MODE 0: the stream goes on (every issue exists; past this tile's last K tile it
        // fetches the NEXT tile's first K tiles: kbias); MODE 1: the workgroup's last tile drains.  RLX: bit g set = phase g of
        // this K tile may use the relaxed count.  HOOK runs after phase 0 (the switch of the DMA addressing to the next tile).
#define PW3_PHASE_END(gidx)                                                                         \
    if (rem == 0) {                                                                                 \
        constexpr int pp_ = ((gidx) + 6) & 3;                                                       \
        const int kk_ = kt + (((gidx) + 6) >> 2);                                                   \
        if (pp_ == 0) issue(0, kk_); else if (pp_ == 1) issue(3, kk_); else if (pp_ == 2) issue(1, kk_); else issue(2, kk_ + 1); \
        if (((rlx >> (gidx)) & 1) && relaxed) wait_relaxed();                                       \
        else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");                                      \
        __builtin_amdgcn_s_barrier();                                                               \
    } else {                                                                                        \
        /* the tile's last two K tiles: `left_` phases remain (this one included).  Its own last half-tile goes out in the first */ \
        /* of them; after that the stream fetches the next tile (kbias), or, on the workgroup's last tile, drains with counted waits */ \
        constexpr int left_ = 4 * rem - (gidx);                                                     \
        constexpr int pp_ = ((gidx) + 6) & 3;                                                       \
        const int kk_ = kt + (((gidx) + 6) >> 2) - kbias;                                           \
        if (left_ > 7 || pf) {                                                                      \
            if (pp_ == 0) issue(0, kk_); else if (pp_ == 1) issue(3, kk_); else if (pp_ == 2) issue(1, kk_); else issue(2, kk_ + 1); \
        }                                                                                           \
        if (pf) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");                                   \
        else wait_left((left_ < 8 ? left_ : 8) - 3);                                                \
        __builtin_amdgcn_s_barrier();                                                               \
    }
        // (X3: terms hi.hi, hi.lo, lo.hi of a fragment pair — [0] = hi, [1] = lo — term-major like the plain kernel's k steps)
#define PW3_MFMA(I0, WARR, J0, GUARD)                                                               \
    __builtin_amdgcn_s_setprio(1);                                                                  \
    if (!(ABL & 1) && (GUARD))                                                                      \
    _Pragma("unroll") for (int ks = 0; ks < (X3 ? 3 : 2); ++ks)                                     \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                               \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                           \
                acc16[(I0) + i][(J0) + j] = Half16<MH>::mfma16(WARR[j][X3 ? (ks == 2) : ks], xf[i][X3 ? (ks == 1) : ks], acc16[(I0) + i][(J0) + j]); \
    __builtin_amdgcn_s_setprio(0);                                                                  \
    __builtin_amdgcn_s_barrier();
#define PW3_KTILE(REM_, KT_, WCUR, WNXT, RLX_, HOOK)                                       \
    {                                                                                               \
        constexpr int rem = (REM_);                /* 0: inside the tile; 2, 1: K tiles left, this one included */ \
        constexpr int rlx = (RLX_);                                                                 \
        const int kt = (KT_);                                                                       \
        const char* bb = smem + (kt & 1) * 4 * HT;                                                  \
        /* phase 0: X-lo(kt) x W-lo(kt) */                                                          \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                               \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                        \
                if (!(ABL & 16)) xf[i][ks] = *reinterpret_cast<const bf16x8*>(bb + xoff + i * 2048 + (((ks * 4 + q4) ^ xkey) << 4)); \
        PW3_PHASE_END(0)                                                                            \
        PW3_MFMA(0, WCUR, 0, true)                                                                  \
        HOOK                                                                                        \
        /* phase 1: W-hi(kt); X-lo x W-hi */                                                        \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                               \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                        \
                if (!(ABL & 16)) whi[j][ks] = *reinterpret_cast<const bf16x8*>(bb + 3 * HT + woff + j * 2048 + (((ks * 4 + q4) ^ wkey) << 4)); \
        PW3_PHASE_END(1)                                                                            \
        PW3_MFMA(0, whi, 2, !half)                                                                  \
        /* phase 2: X-hi(kt); X-hi x W-hi */                                                        \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                               \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                        \
                if (!(ABL & 16)) xf[i][ks] = *reinterpret_cast<const bf16x8*>(bb + HT + xoff + i * 2048 + (((ks * 4 + q4) ^ xkey) << 4)); \
        PW3_PHASE_END(2)                                                                            \
        PW3_MFMA(4, whi, 2, !half)                                                                  \
        /* phase 3: W-lo(kt+1) into the other W-lo register set; X-hi x W-lo(kt) */                 \
        {   /* (on the workgroup's last K tile this reads a buffer nobody refilled: unused) */     \
            const char* bn = smem + ((kt + 1) & 1) * 4 * HT + 2 * HT;                               \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                           \
                _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                    \
                    if (!(ABL & 16)) WNXT[j][ks] = *reinterpret_cast<const bf16x8*>(bn + woff + j * 2048 + (((ks * 4 + q4) ^ wkey) << 4)); \
        }                                                                                           \
        PW3_PHASE_END(3)                                                                            \
        PW3_MFMA(4, WCUR, 0, true)                                                                  \
    }
#define PW3_ROLL                                                                                    \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                   \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) wlo[j][ks] = wnx[j][ks];
        // the tile's phases 0 .. 4 wait for half-tiles 3 .. 7 of its first K tiles: the stores of the previous epilogue sit behind them
        PW3_KTILE(0, 0, wlo, wnx, 15, )
        PW3_ROLL
        PW3_KTILE(0, 1, wlo, wnx, 1, )
        PW3_ROLL
        int kt0 = 2;
        for (; kt0 + 2 < nkt; ++kt0) {              // steady state: every issue exists, five half-tiles stay in flight
            PW3_KTILE(0, kt0, wlo, wnx, 0, )
            PW3_ROLL
        }
        // the stream runs on: after X-hi of this tile's last K tile (phase 0 of K tile nkt - 2) every issue fetches the next tile
        // (nkt is even: buffer parities continue), and the last phase leaves its W-lo(0) in wlo
        PW3_KTILE(2, kt0, wlo, wnx, 0,
                  if (pf) { set_src(tm_n * 256, n0_n); issue_consts(n0_n, par ^ 1); kbias = nkt; })
        PW3_ROLL
        ++kt0;
        PW3_KTILE(1, kt0, wlo, wnx, 0, )
        PW3_ROLL
        kbias = 0;
#undef PW3_KTILE
#undef PW3_PHASE_END
#undef PW3_MFMA
#undef PW3_ROLL
        if (wm == 0) __builtin_amdgcn_s_barrier();      // even out the barrier count: both groups take the epilogue together
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);              // the stores below stay BEHIND the K loop's DMAs in the wave's queue (vmcnt bookkeeping)
        PW3_STAMP(1)

        // ---- epilogue from the accumulators ---------------------------------------------------------------------------
        const int lane_e = lane_now();
        const int r16e = lane_e & 15, q4e = lane_e >> 4;
        if (R2) {
            // Res2Net step: y = BN(ReLU(.)) in the S32 layout into the chain output; U_next = y + c_next (fp32 chunk of the tdnn1
            // output) in the S32 layout into the next step's input.  Only the wave columns that hold channels store anything.
            // c_next comes through the LDS ring, which is free here because this form does not run its DMA stream on into the next
            // tile: 256 rows x 512 bytes by LDS-DMA from all eight waves, one wait.  (Loaded straight into registers, four at a
            // time between the stores, every wait for a load also waited for the acknowledgement of every store issued before it:
            // the epilogue took 52 k cycles against 46 k for the K loop.)
            __builtin_amdgcn_s_barrier();                       // (group 1 is past its last fragment read)
            // The ring holds one 256-row image of 512-byte rows at a time: first the c tile (fp32), which the valid waves turn IN
            // PLACE into U = y + c in the S32 layout (a 128-byte block of a row — 32 channels as fp32, or their hi | lo planes —
            // is read and written by the four lanes of ONE wave, and a wave's LDS operations execute in order), then y.  Each
            // image leaves through all eight waves as whole rows: one wave instruction = two rows = 1 KiB contiguous.  (Stored
            // straight from the accumulator layout an instruction wrote 64 separate 16-byte pieces and the epilogue was bound by
            // the CU's store path: 35.6 k cycles per tile.)  16-byte chunk k of a block sits at k ^ (row & 7): two-way bank
            // conflicts at most for the accumulator-layout accesses, none for the row copies.
            const float* Cn = reinterpret_cast<const float*>(p.R);
            char* Ub = reinterpret_cast<char*>(p.Y2);
            const int nchunk = p.N >> 2;                        // 16-byte chunks per image row (32 or 16)
            auto copy_out = [&](char* dst, int64_t ld_bytes) {
                const int lane_c = lane_now();
#pragma unroll 4
                for (int q = 0; q < 16; ++q) {
                    const int pidx = (q * 8 + wave) * 64 + lane_c;
                    const int row = pidx >> 5, ch = pidx & 31;
                    const u32x4 v = *reinterpret_cast<const u32x4*>(smem + row * 512 + (ch & ~7) * 16 + (((ch & 7) ^ (row & 7)) << 4));
                    if (m0 + row < p.M && ch < nchunk && !(ABL & 8)) *reinterpret_cast<u32x4*>(dst + (int64_t)(m0 + row) * ld_bytes + ch * 16) = v;
                }
            };
            // a lane's four values -> 8 bytes of the hi plane and 8 bytes of the lo plane of its block
            auto put_s32 = [&](const f32x4& v, int ml, int nl) {
                uint32_t hd[2], ld[2];
#pragma unroll
                for (int d = 0; d < 2; ++d) x3_split2(v[2 * d], v[2 * d + 1], hd[d], ld[d]);
                const int kh = (nl & 31) >> 3;                  // chunk of the hi plane holding channels nl .. nl + 3 (lo: + 4)
                char* blk = smem + ml * 512 + (nl >> 5) * 128 + (nl & 4) * 2;
                *reinterpret_cast<uint2*>(blk + ((kh ^ (ml & 7)) << 4)) = make_uint2(hd[0], hd[1]);
                *reinterpret_cast<uint2*>(blk + (((kh + 4) ^ (ml & 7)) << 4)) = make_uint2(ld[0], ld[1]);
            };
            if (Cn) {
                const int lane_c = lane_now();
#pragma unroll 4
                for (int q = 0; q < 16; ++q) {
                    const int pidx = (q * 8 + wave) * 64 + lane_c;      // 16-byte chunk position of the 256 x 32-chunk image
                    const int row = pidx >> 5, pos = pidx & 31;
                    const int cs = min((pos & ~7) | ((pos ^ row) & 7), nchunk - 1);
                    const int m = min(m0 + row, p.M - 1);
                    const float* src = Cn + (int64_t)m * p.ldr + (cs << 2);
                    __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(smem + (q * 8 + wave) * 1024), 16, 0, 0);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            const bool valid_cols = wn * 64 < p.N;
            if (valid_cols) {
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
                    f32x4 sc[2], sh[2];
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int nl = wn * 64 + (2 * jp + jj) * 16 + 4 * q4e;
                        sc[jj] = *reinterpret_cast<const f32x4*>(cb + 1024 + nl * 4);
                        sh[jj] = *reinterpret_cast<const f32x4*>(cb + 2048 + nl * 4);
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int ml = wm * 128 + i * 16 + r16e;
                        f32x4 cn[2];
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) {
                            const int j = 2 * jp + jj;
                            const int nl = wn * 64 + j * 16 + 4 * q4e;
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc16[i][j][e] = fmaf(fmaxf(acc16[i][j][e], 0.0f), sc[jj][e], sh[jj][e]);
                            if (Cn) cn[jj] = *reinterpret_cast<const f32x4*>(smem + ml * 512 + (nl >> 5) * 128 + ((((nl & 31) >> 2) ^ (ml & 7)) << 4));
                        }
                        if (Cn) {
#pragma unroll
                            for (int jj = 0; jj < 2; ++jj) put_s32(acc16[i][2 * jp + jj] + cn[jj], ml, wn * 64 + (2 * jp + jj) * 16 + 4 * q4e);
                        }
                    }
                }
            }
            if (Cn) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                   // the U image is complete
                copy_out(Ub, (int64_t)p.lda2 * 4);
                __builtin_amdgcn_s_barrier();                   // ... and has been read
            }
            if (valid_cols) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 8; ++i) put_s32(acc16[i][j], wm * 128 + i * 16 + r16e, wn * 64 + j * 16 + 4 * q4e);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            copy_out(reinterpret_cast<char*>(p.Y), (int64_t)p.ldy * 4);
            if (more) {         // the next tile starts from scratch, like the first one
                __builtin_amdgcn_s_barrier();                   // every wave has read its c values: the ring may be refilled
                set_src(tm_n * 256, n0_n);
                issue_consts(n0_n, par ^ 1);
                issue_prologue();
                wait_left(5);
                __builtin_amdgcn_s_barrier();
                const int lane_w = lane_now();
                const int r16w = lane_w & 15, q4w = lane_w >> 4;
                const int woffw = (wn * 32 + r16w) * 128, wkeyw = ((wn * 32 + r16w) >> 1) & 7;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) wlo[j][ks] = *reinterpret_cast<const bf16x8*>(smem + 2 * HT + woffw + j * 2048 + (((ks * 4 + q4w) ^ wkeyw) << 4));
            }
        } else if (X3) {
            // fp32 out: a lane's accumulator register group IS 16 contiguous bytes (4 channels of one frame)
            float* Yf = reinterpret_cast<float*>(p.Y);
            // side outputs (GemmParams::side_a / side_b): this wave's 64 channels, when they lie in the first two chunks of the
            // output, leave in the S32 layout instead of fp32 — hi / lo halves of four values, then one v_permlane16_swap per dword
            // with the lane 16 up, so that the even lane of a pair holds the hi halves of eight consecutive channels and the odd
            // lane their lo halves: 16-byte stores
            const int sc_w = p.side_c;
            // wave-uniform; 0: side_a (or, with y_s32, the whole output: Y itself is an S32 buffer), 1: side_b, else fp32
            const int noff = hsel == 2 ? 32 : 0;       // a second column half: this wave's channels start 32 past its whole-tile origin
            const int side_chunk = p.y_s32 ? 0 : (sc_w > 0 && tn == 0) ? (wn * 64) / sc_w : 2;
            char* sbase = p.y_s32 ? reinterpret_cast<char*>(p.Y) : side_chunk == 0 ? reinterpret_cast<char*>(p.side_a) : reinterpret_cast<char*>(p.side_b);
            const int64_t sld = (int64_t)(p.y_s32 ? p.ldy : side_chunk == 0 ? p.side_lda : p.side_ldb) * 4;
            const int scol0 = p.y_s32 ? n0 + wn * 64 : side_chunk < 2 ? wn * 64 + noff - side_chunk * sc_w : 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (half && j >= 2) break;          // a column half: the wave's first 32 channels only
                const int nl = wn * 64 + j * 16 + 4 * q4e;
                const f32x4 sc = *reinterpret_cast<const f32x4*>(cb + 1024 + nl * 4);
                const f32x4 sh = *reinterpret_cast<const f32x4*>(cb + 2048 + nl * 4);
                float* yl = Yf + (int64_t)(m0 + wm * 128 + r16e) * p.ldy + n0 + nl;
                const int na = (scol0 + j * 16 + 4 * q4e) & ~7;                           // the lane pair's first channel inside the side buffer
                const int boff = (na >> 5) * 128 + (na & 31) * 2 + (q4e & 1) * 64;       // even q4: hi plane, odd q4: lo plane
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = acc16[i][j][e];
                        if (X3 && CV) t *= inv_in;
                        if (!(ABL & 4)) {
                            t = apply_act(t, EPI == EPI_GELU ? ACT_GELU : EPI == EPI_RELU ? ACT_RELU : ACT_NONE);     // nn.GELU() exact form
                            t = fmaf(t, sc[e], sh[e]);
                        }
                        v[e] = t;
                    }
                    if (CS) acc16[i][j] = v;
                    const int m = m0 + wm * 128 + i * 16 + r16e;
                    const bool in = m < p.M && !(ABL & 8);
                    if (side_chunk < 2) {
                        uint32_t hd[2], ld[2];
#pragma unroll
                        for (int d = 0; d < 2; ++d) x3_split2(v[2 * d], v[2 * d + 1], hd[d], ld[d]);
                        const auto s0 = __builtin_amdgcn_permlane16_swap(hd[0], ld[0], false, false);
                        const auto s1 = __builtin_amdgcn_permlane16_swap(hd[1], ld[1], false, false);
                        if (in) *reinterpret_cast<u32x4*>(sbase + (int64_t)m * sld + boff) = u32x4{s0[0], s1[0], s0[1], s1[1]};
                    } else if (in) {
                        *reinterpret_cast<f32x4*>(yl + (int64_t)i * 16 * p.ldy) = v;
                    }
                }
            }
        } else {
            char* Yb = reinterpret_cast<char*>(p.Y);
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                // a column half: the wave's first 32 channels only (the sums below are formed for all four j; the stores of j >= 2 are masked)
                if (half && jp == 1) break;
                f32x4 sc[2], sh[2];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int nl = wn * 64 + (2 * jp + jj) * 16 + 4 * q4e;
                    sc[jj] = *reinterpret_cast<const f32x4*>(cb + 1024 + nl * 4);
                    sh[jj] = *reinterpret_cast<const f32x4*>(cb + 2048 + nl * 4);
                }
                // after the swaps lane (q4e, r16e) holds channels c0 .. c0 + 7 of frame r16e (+ 16 i): 16 contiguous bytes
                const int c0 = n0 + wn * 64 + jp * 32 + (q4e & 1) * 16 + (q4e >> 1) * 8;
                char* yl = Yb + ((int64_t)(m0 + wm * 128 + r16e) * p.ldy + c0) * 2;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float v0[4], v1[4];
                    if (ABL & 4) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v0[e] = acc16[i][2 * jp][e]; v1[e] = acc16[i][2 * jp + 1][e]; }
                    } else {
                        act4<EPI>(v0, acc16[i][2 * jp], sc[0], sh[0]);
                        act4<EPI>(v1, acc16[i][2 * jp + 1], sc[1], sh[1]);
                    }
                    if (CS) {           // the column sums below take the activated fp32 values
                        acc16[i][2 * jp] = f32x4{v0[0], v0[1], v0[2], v0[3]};
                        acc16[i][2 * jp + 1] = f32x4{v1[0], v1[1], v1[2], v1[3]};
                    }
                    const auto s0 = __builtin_amdgcn_permlane16_swap(Half16<H>::pack2(v0[0], v0[1]), Half16<H>::pack2(v1[0], v1[1]), false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(Half16<H>::pack2(v0[2], v0[3]), Half16<H>::pack2(v1[2], v1[3]), false, false);
                    const int m = m0 + wm * 128 + i * 16 + r16e;
                    if (m < p.M && !(ABL & 8))
                        *reinterpret_cast<u32x4*>(yl + (int64_t)i * 16 * p.ldy * 2) = u32x4{s0[0], s1[0], s0[1], s1[1]};
                }
            }
        }
        if (CS) {
            // per-utterance column sums of this wave's 128 frames x 64 channels:
            //   colsum[((tm*2 + wm)*2 + seg) * N + n], seg 0 = the utterance of the tile's first row, seg 1 = the next one
            const int lo = wm * 128;
            const int rb = (m0 / p.T + 1) * p.T - m0;         // first tile row that belongs to the next utterance
            const int rend = min(256, p.M - m0);
            const bool whole = (lo + 128 <= rend) && (lo + 128 <= rb || lo >= rb);      // wave-uniform: one segment, every row valid
            float* csp = p.colsum + ((int64_t)(tm * 2 + wm) * 2) * p.N + n0 + wn * 64 + 4 * q4e;
            const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
            const int jn = half ? 2 : 4;            // a column half owns the sums of its first 32 channels only (the other half item writes the rest)
            // (the zeros that are STORED are made per use: hoisted out of the tile loop they were spilled, and every reload of a spill
            //  is a vector-memory load whose wait also waits for the tile's stores)
            auto fresh_zero4 = [&]() { float z = 0.f; asm volatile("" : "+v"(z)); return f32x4{z, z, z, z}; };
            // (sums and sums of squares in separate passes over the accumulators: one set of partial sums live at a time)
            if (whole) {
                const int sg = lo >= rb ? 1 : 0;
#pragma unroll
                for (int kind = 0; kind < CS; ++kind) {
                    float* cs = csp + (kind ? p.colsum_stride : 0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        f32x4 s = zero4;
                        if (!(ABL & 64)) {
#pragma unroll
                            for (int i = 0; i < 8; ++i) s += kind ? acc16[i][j] * acc16[i][j] : acc16[i][j];
#pragma unroll
                            for (int e = 0; e < 4; ++e) s[e] = row16_sum(s[e]);
                        }
                        if (r16e == 0 && j < jn) {
                            *reinterpret_cast<f32x4*>(cs + j * 16 + (int64_t)sg * p.N) = s;
                            *reinterpret_cast<f32x4*>(cs + j * 16 + (int64_t)(1 - sg) * p.N) = fresh_zero4();
                        }
                    }
                }
            } else if (lo + 128 <= rend) {
                // an utterance boundary inside this wave's 128 valid rows: total and second-segment sums (one weight per 16-row
                // block, set up once), first segment = total - second (fp32: the difference carries ~1e-7 of the total)
                float w1[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) w1[i] = (lo + i * 16 + r16e >= rb) ? 1.0f : 0.0f;
#pragma unroll
                for (int kind = 0; kind < CS; ++kind) {
                    float* cs = csp + (kind ? p.colsum_stride : 0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        f32x4 sa = zero4, s1 = zero4;
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const f32x4 v = kind ? acc16[i][j] * acc16[i][j] : acc16[i][j];
                            sa += v; s1 += v * w1[i];
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) { sa[e] = row16_sum(sa[e]); s1[e] = row16_sum(s1[e]); }
                        if (r16e == 0 && j < jn) {
                            *reinterpret_cast<f32x4*>(cs + j * 16) = sa - s1;
                            *reinterpret_cast<f32x4*>(cs + j * 16 + p.N) = s1;
                        }
                    }
                }
            } else {
                // rows past M (the last M-tile only): every row weighted on its own
#pragma unroll
                for (int kind = 0; kind < CS; ++kind) {
                    float* cs = csp + (kind ? p.colsum_stride : 0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        f32x4 s0 = zero4, s1 = zero4;
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const int row = lo + i * 16 + r16e;
                            const float w0 = (row < rb && row < rend) ? 1.0f : 0.0f;
                            const float w1 = (row >= rb && row < rend) ? 1.0f : 0.0f;
                            const f32x4 v = kind ? acc16[i][j] * acc16[i][j] : acc16[i][j];
                            s0 += v * w0; s1 += v * w1;
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) { s0[e] = row16_sum(s0[e]); s1[e] = row16_sum(s1[e]); }
                        if (r16e == 0 && j < jn) {
                            *reinterpret_cast<f32x4*>(cs + j * 16) = s0;
                            *reinterpret_cast<f32x4*>(cs + j * 16 + p.N) = s1;
                        }
                    }
                }
            }
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        PW3_STAMP(3)
        ++ntile_done;
        if (!more) break;
        // every row of the tile just stored was inside M: each wave issued exactly NST stores behind the K loop's DMAs
        relaxed = !R2 && !half && !(ABL & 128) && !(ABL & 8) && (m0 + 256 <= p.M);       // (R2, half items: another store count)
        w = w_next; tm = tm_n; tn = tn_n; hsel = hsel_n;
    }
    if (DBG3 && (p.debug & 16384) && p.ts) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long tend = __builtin_readcyclecounter();
        if ((tid & 63) == 0) {          // one record per wave: [workgroup][wave][8]
            unsigned long long* o = reinterpret_cast<unsigned long long*>(p.ts) + ((int64_t)blockIdx.x * 8 + wave) * 8;
            for (int i = 0; i < 4; ++i) o[i] = tacc[i];
            o[4] = (unsigned long long)ntile_done; o[5] = tend - tstart;
            unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            o[6] = tstart; o[7] = xcc;
        }
    }
#undef PW3_STAMP
}

// workgroups of a persistent launch: the grid cap, or one per tile — or, for the 16-bit pointwise form with the tail split on, one per
// column half when the tiles fill at most half of the capped grid (small batches: 128 tiles of a K = 1024 layer at B = 20 become 256 halves)
inline int pw3_grid(const GemmParams& p, int ntiles, bool splittable) {
    const int cap = pw3_grid_cap(p);
    if (ntiles >= cap) return cap;
    return (splittable && p.tail_split && 2 * ntiles <= cap) ? 2 * ntiles : ntiles;
}

template <int EPI, int CS, bool X3>
hipError_t launch_inst(const GemmParams& p, hipStream_t stream) {
    const int ntiles = ((p.M + 255) / 256) * (p.N / 256);
    static DeviceOnce attr;
    if (hipError_t e = set_max_dynamic_lds(attr, reinterpret_cast<const void*>(gemm_pw3_kernel<EPI, CS, X3>), PW3_LDS)) return e;
    const int grid = pw3_grid(p, ntiles, true);
    hipLaunchKernelGGL((gemm_pw3_kernel<EPI, CS, X3>), dim3(grid), dim3(512), PW3_LDS, stream, p);
    return hipGetLastError();
}

// the kernel forms its per-lane byte offsets with 24-bit multiplies and 32-bit sums
inline bool pw3_offsets_fit(const GemmParams& p, int esz) {
    return (int64_t)p.M + 512 < (1 << 24) && (int64_t)p.lda * esz < (1 << 24) && (int64_t)p.Kp * esz < (1 << 24) && (int64_t)p.Wrows < (1 << 24) &&
           ((int64_t)p.M + 512) * p.lda * esz < ((int64_t)1 << 32) && (int64_t)p.Wrows * p.Kp * esz < ((int64_t)1 << 32);
}

inline uint32_t t_magic_of(int T) { return (uint32_t)(0xFFFFFFFFull / (uint32_t)T) + 1u; }       // ceil(2^32 / T), T >= 2

hipError_t launch_cv(const GemmParams& p_in, hipStream_t stream) {
    GemmParams p = p_in;
    p.t_magic = t_magic_of(p.T);
    const int ntiles = ((p.M + 255) / 256) * (p.N / 256);
    static DeviceOnce attr;
    if (hipError_t e = set_max_dynamic_lds(attr, reinterpret_cast<const void*>(gemm_pw3_kernel<EPI_GELU, 0, true, false, true>), PW3_LDS)) return e;
    const int cap = pw3_grid_cap(p);
    hipLaunchKernelGGL((gemm_pw3_kernel<EPI_GELU, 0, true, false, true>), dim3(ntiles < cap ? ntiles : cap), dim3(512), PW3_LDS, stream, p);
    return hipGetLastError();
}

template <int EPI, bool X3>
hipError_t launch_cs(const GemmParams& p, hipStream_t stream) {
    if (!p.colsum) return launch_inst<EPI, 0, X3>(p, stream);
    return p.colsum_sq ? launch_inst<EPI, 2, X3>(p, stream) : launch_inst<EPI, 1, X3>(p, stream);
}

}  // namespace

template <int EPI, typename H>
hipError_t launch_cv16_inst(const GemmParams& p_in, hipStream_t stream) {
    GemmParams p = p_in;
    p.t_magic = t_magic_of(p.T);
    const int ntiles = ((p.M + 255) / 256) * (p.N / 256);
    static DeviceOnce attr;
    if (hipError_t e = set_max_dynamic_lds(attr, reinterpret_cast<const void*>(gemm_pw3_kernel<EPI, 0, false, false, true, H>), PW3_LDS)) return e;
    const int cap = pw3_grid_cap(p);
    hipLaunchKernelGGL((gemm_pw3_kernel<EPI, 0, false, false, true, H>), dim3(ntiles < cap ? ntiles : cap), dim3(512), PW3_LDS, stream, p);
    return hipGetLastError();
}

// The persistent kernel takes the plain pointwise layers of the 256 x 256 kernel's contract (no conv-gather, no residual) with
// whole N tiles, at least four K tiles, all three per-channel vectors present, and more tiles than CUs (below that a persistent
// workgroup has no second tile to overlap anything with).
bool gemm_pw3_supported(const GemmParams& p, bool bf16) {
    if (!gemm_pw2_supported(p, bf16)) return false;
    if (p.taps > 1 || p.R || p.A3 || p.f16) return false;
    if (!pw3_offsets_fit(p, 2)) return false;
    if (p.N % 256 != 0 || p.Kp < 256) return false;
    if (!p.bias || !p.scale || !p.shift) return false;
    if (p.act2 != ACT_NONE || !(p.act1 == ACT_NONE || p.act1 == ACT_RELU || p.act1 == ACT_GELU)) return false;
    if (p.num_cu <= 0 || p.num_cu > 1024) return false;
    const int ntiles = ((p.M + 255) / 256) * (p.N / 256);
    // more tiles than workgroups (something to overlap an epilogue with), or so few that every tile can be walked as two column halves
    return ntiles > pw3_grid_cap(p) || (p.tail_split && 2 * ntiles <= pw3_grid_cap(p) && pw3_grid_cap(p) <= p.num_cu);
}

// developer option pw3_cus = n (GemmParams::pw3_cus, from the handle: SVHIP_PW3_CUS at svhip_create or svhip_set_option): launch at
// most n workgroups, so that a small test problem walks several tiles per workgroup (relaxed / strict waits, constant strips of both
// parities, masked last tile); 0 disables the persistent kernels
int pw3_grid_cap(const GemmParams& p) {
    if (p.pw3_cus == 0) return 1 << 30;
    return p.pw3_cus > 0 ? (p.pw3_cus < p.num_cu ? p.pw3_cus : p.num_cu) : p.num_cu;
}

hipError_t launch_gemm_pw3(const GemmParams& p, hipStream_t stream) {
    if (!gemm_pw3_supported(p, true) || p.M <= 0 || p.Wrows < p.N) return hipErrorInvalidValue;
    switch (p.act1) {
        case ACT_NONE: return launch_cs<EPI_NONE, false>(p, stream);
        case ACT_RELU: return launch_cs<EPI_RELU, false>(p, stream);
        case ACT_GELU: return launch_cs<EPI_GELU, false>(p, stream);
        default: return hipErrorInvalidValue;
    }
}

// The X3 form (p.x3 == 2: A and W in the S32 split layout, fp32 out): the GELU layers of SVHIP_F32X3 handles
bool gemm_pw3x3_supported(const GemmParams& p) {
    if (p.x3 != 2 || p.out_f32 || p.bias_utt || p.A2 || p.A3 || p.R || p.taps > 1 || p.y_s32) return false;
    if (!pw3_offsets_fit(p, 4)) return false;
    if (p.act1 != ACT_GELU || p.act2 != ACT_NONE) return false;
    if (!p.bias || !p.scale || !p.shift) return false;
    if (p.N % 256 != 0 || p.K != p.Kp || p.K % 64 != 0 || p.K < 128 || p.lda < p.K || p.lda % 32 != 0 || p.ldy % 4 != 0) return false;      // whole 32-k blocks, an even number of them
    if (p.colsum && (p.T < 256 || p.M % p.T != 0)) return false;
    if (p.side_c && (!(p.side_c == 64 || p.side_c == 128) || !p.side_a || !p.side_b || p.side_lda % 32 != 0 || p.side_ldb % 32 != 0 || p.side_lda < p.side_c ||
                     p.side_ldb < p.side_c || ((reinterpret_cast<uintptr_t>(p.side_a) | reinterpret_cast<uintptr_t>(p.side_b)) & 127)))
        return false;
    if ((reinterpret_cast<uintptr_t>(p.A) | reinterpret_cast<uintptr_t>(p.W) | reinterpret_cast<uintptr_t>(p.Y) | reinterpret_cast<uintptr_t>(p.bias) |
         reinterpret_cast<uintptr_t>(p.scale) | reinterpret_cast<uintptr_t>(p.shift)) & 15) return false;
    if (p.num_cu <= 0 || p.num_cu > 1024 || p.M <= 0 || p.Wrows < p.N) return false;
    // (any tile count: unlike the 16-bit form there is no per-tile twin to hand a small grid to, and the generic split-in-registers
    //  kernels it would fall back to are 3 - 4 x slower than one tile per workgroup of this one)
    return p.pw3_cus != 0;
}

// One step of a Res2Net chain on F32X3 handles (the R2 form): A = U_j (M, cin) S32, W = the step's conv weight (cin, 3 cin) S32
// (k = tap * cin + c), Y = chain output in S32 (row stride ldy elements, the chunk's column offset in the pointer), and with
// R / Y2: R = the next chunk of the fp32 tdnn1 output (row stride ldr), Y2 = U_{j+1} (M, lda2) S32.
bool gemm_pw3r2_supported(const GemmParams& p) {
    if (p.x3 != 2 || p.taps != 3 || p.A2 || p.A3 || p.bias_utt || p.colsum || p.out_f32) return false;
    if (!pw3_offsets_fit(p, 4)) return false;
    if (!(p.cin == 64 || p.cin == 128) || p.N != p.cin || p.K != 3 * p.cin || p.Kp != p.K) return false;
    if (p.act1 != ACT_RELU || p.act2 != ACT_NONE || p.pad_mode != PAD_REFLECT) return false;
    if (!p.bias || !p.scale || !p.shift || !p.Y) return false;
    if ((p.R == nullptr) != (p.Y2 == nullptr)) return false;
    if (p.lda < p.cin || p.lda % 32 != 0 || p.ldy % 32 != 0 || (p.Y2 && (p.lda2 % 32 != 0 || p.ldr % 4 != 0))) return false;
    if (p.T <= 2 * p.dil || p.dil < 1 || p.dil > 4 || p.T >= 65536 || p.M <= 0 || p.M % p.T != 0 || p.Wrows < p.N) return false;
    if ((reinterpret_cast<uintptr_t>(p.A) | reinterpret_cast<uintptr_t>(p.W) | reinterpret_cast<uintptr_t>(p.Y) | reinterpret_cast<uintptr_t>(p.Y2) |
         reinterpret_cast<uintptr_t>(p.R) | reinterpret_cast<uintptr_t>(p.bias) | reinterpret_cast<uintptr_t>(p.scale) | reinterpret_cast<uintptr_t>(p.shift)) & 15) return false;
    if (p.num_cu <= 0 || p.num_cu > 1024) return false;
    return p.pw3_cus != 0;
}

hipError_t launch_gemm_pw3r2(const GemmParams& p_in, hipStream_t stream) {
    if (!gemm_pw3r2_supported(p_in)) return hipErrorInvalidValue;
    GemmParams p = p_in;
    p.t_magic = t_magic_of(p.T);
    const int ntiles = (p.M + 255) / 256;
    static DeviceOnce attr;
    if (hipError_t e = set_max_dynamic_lds(attr, reinterpret_cast<const void*>(gemm_pw3_kernel<EPI_RELU, 0, true, true>), PW3_LDS)) return e;
    const int cap = pw3_grid_cap(p);
    hipLaunchKernelGGL((gemm_pw3_kernel<EPI_RELU, 0, true, true>), dim3(ntiles < cap ? ntiles : cap), dim3(512), PW3_LDS, stream, p);
    return hipGetLastError();
}

hipError_t launch_gemm_pw3x3(const GemmParams& p, hipStream_t stream) {
    if (!gemm_pw3x3_supported(p)) return hipErrorInvalidValue;
    return launch_cs<EPI_GELU, true>(p, stream);
}

// The conv-gather X3 form (ECAPA blocks.0 on F32X3 handles): A = the layer input in the S32 layout with rows of p.cin channels
// (a multiple of 32: the real channels zero-padded), W = [N][Kp] S32 with k = tap * cin + c and Kp = taps * cin rounded up to a
// multiple of 64 (zero columns), reflect padding inside each utterance, GELU -> BN, fp32 out.
bool gemm_pw3cv_supported(const GemmParams& p) {
    if (p.x3 != 2 || p.out_f32 || p.bias_utt || p.A2 || p.A3 || p.R || p.colsum || p.side_c) return false;
    if (!pw3_offsets_fit(p, 4)) return false;
    if (p.taps < 3 || p.taps > 7 || !(p.taps & 1) || p.pad_mode != PAD_REFLECT || p.dil < 1 || (p.taps >> 1) * p.dil > 8) return false;
    if (p.act1 != ACT_GELU || p.act2 != ACT_NONE || !p.bias || !p.scale || !p.shift) return false;
    if (p.cin % 32 != 0 || p.cin < 32 || p.lda != p.cin || p.K != p.taps * p.cin || p.Kp % 64 != 0 || p.Kp < p.K || p.Kp - p.K >= 64 || p.Kp > 63 * 32) return false;
    if (p.N % 256 != 0 || p.ldy % 4 != 0 || p.Wrows < p.N || (p.y_s32 && (p.ldy % 32 != 0 || (reinterpret_cast<uintptr_t>(p.Y) & 127)))) return false;
    if (p.T <= (p.taps >> 1) * p.dil || p.T >= 65536 || p.M <= 0 || p.M % p.T != 0) return false;
    if ((reinterpret_cast<uintptr_t>(p.A) | reinterpret_cast<uintptr_t>(p.W) | reinterpret_cast<uintptr_t>(p.Y) | reinterpret_cast<uintptr_t>(p.bias) |
         reinterpret_cast<uintptr_t>(p.scale) | reinterpret_cast<uintptr_t>(p.shift)) & 15) return false;
    if (p.num_cu <= 0 || p.num_cu > 1024) return false;
    return p.pw3_cus != 0;
}

hipError_t launch_gemm_pw3cv(const GemmParams& p, hipStream_t stream) {
    if (!gemm_pw3cv_supported(p)) return hipErrorInvalidValue;
    return launch_cv(p, stream);
}

// The conv-gather form on 16-bit operands (bf16, or fp16 with p.f16): odd taps, cin % 64 == 0 (a 128-byte K tile = 64 channels of one
// tap), reflect or zero padding inside each utterance, epilogues GELU -> BN (ECAPA blocks.0), BN -> LeakyReLU(0.3) (RawNet2 conv1), none
// (conv2).  The kernel reads its three
// per-channel vectors by DMA: absent ones are replaced by the caller's constant vectors (GemmParams::zeros / ones, >= N floats).
bool gemm_pw3cv16_supported(const GemmParams& p) {
    if (p.x3 || p.out_f32 || p.bias_utt || p.A2 || p.A3 || p.R || p.colsum || p.side_c || p.y_s32) return false;
    if (!pw3_offsets_fit(p, 2)) return false;
    if (p.taps < 3 || p.taps > 7 || !(p.taps & 1) || p.dil < 1 || (p.taps >> 1) * p.dil > 8) return false;
    if (!(p.pad_mode == PAD_REFLECT || (p.pad_mode == PAD_ZERO && p.zero_page))) return false;
    if (p.pad_mode == PAD_ZERO) {      // the zero page is addressed as a 32-bit offset from any K tile base of A: behind A's rows, within 4 GiB
        const uintptr_t a0 = reinterpret_cast<uintptr_t>(p.A), z = reinterpret_cast<uintptr_t>(p.zero_page);
        if (z < a0 + (uintptr_t)p.M * p.lda * 2 || z - a0 >= ((uintptr_t)1 << 32) - (uintptr_t)16 * p.lda * 2 - 4096) return false;
    }
    const bool gelu = p.act1 == ACT_GELU && p.act2 == ACT_NONE, bnlr = p.act1 == ACT_NONE && p.act2 == ACT_LRELU03, none = p.act1 == ACT_NONE && p.act2 == ACT_NONE;
    if (!(gelu || bnlr || none) || (p.f16 && gelu)) return false;
    if (!(p.bias || p.zeros) || !(p.scale || p.ones) || !(p.shift || p.zeros) || ((p.scale == nullptr) != (p.shift == nullptr))) return false;
    if (p.cin % 64 != 0 || p.cin < 64 || p.lda < p.cin || p.lda % 8 != 0 || p.K != p.taps * p.cin) return false;
    if (p.Kp != p.K) return false;
    if (p.Kp % 128 != 0 || p.Kp < 256 || p.Kp > 63 * 64) return false;             // an even number (>= 4) of 64-wide K tiles, kt < 64
    if (p.N % 256 != 0 || p.ldy % 8 != 0 || p.Wrows < p.N) return false;
    if (p.T <= (p.taps >> 1) * p.dil || p.T >= 65536 || p.M <= 0 || p.M % p.T != 0) return false;
    if ((int64_t)(p.M + 16) * p.lda * 2 >= (int64_t)1 << 31) return false;         // 32-bit per-lane byte offsets
    if ((reinterpret_cast<uintptr_t>(p.A) | reinterpret_cast<uintptr_t>(p.W) | reinterpret_cast<uintptr_t>(p.Y) |
         reinterpret_cast<uintptr_t>(p.bias) | reinterpret_cast<uintptr_t>(p.scale) | reinterpret_cast<uintptr_t>(p.shift) |
         reinterpret_cast<uintptr_t>(p.zeros) | reinterpret_cast<uintptr_t>(p.ones) | reinterpret_cast<uintptr_t>(p.zero_page)) & 15) return false;
    if (p.num_cu <= 0 || p.num_cu > 1024) return false;
    return ((p.M + 255) / 256) * (p.N / 256) > pw3_grid_cap(p);
}

hipError_t launch_gemm_pw3cv16(const GemmParams& p_in, hipStream_t stream) {
    if (!gemm_pw3cv16_supported(p_in)) return hipErrorInvalidValue;
    GemmParams p = p_in;
    if (!p.bias) p.bias = p.zeros;
    if (!p.scale) { p.scale = p.ones; p.shift = p.zeros; }
    if (p.act2 == ACT_LRELU03) return p.f16 ? launch_cv16_inst<EPI_BN_LRELU03, f16_t>(p, stream) : launch_cv16_inst<EPI_BN_LRELU03, bf16_t>(p, stream);
    if (p.act1 == ACT_GELU) return launch_cv16_inst<EPI_GELU, bf16_t>(p, stream);
    return p.f16 ? launch_cv16_inst<EPI_NONE, f16_t>(p, stream) : launch_cv16_inst<EPI_NONE, bf16_t>(p, stream);
}

}  // namespace svhip
