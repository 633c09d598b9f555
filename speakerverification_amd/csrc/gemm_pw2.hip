// gemm_pw2.hip — bf16 pointwise / dilated-conv GEMM, 256 x 256 tile, role-staggered wave groups (gfx950).
//
// Same contract as gemm_pw.hip (Y = epi(A . W^T), bf16 in / fp32 accumulate / bf16 out), built for the big layers
// (blocks.0, tdnn1 / tdnn2, mfa).  Structure:
//
//   * 512 threads = 8 waves, 2 (M) x 4 (N), 128 frames x 64 channels per wave, v_mfma_f32_16x16x32_bf16 with the weights as
//     the MFMA A operand, so a lane owns 4 consecutive channels of one frame (packed 8-byte epilogue).
//   * K advances in 64-wide K tiles of FOUR PHASES, one accumulator quadrant (64 frames x 32 channels, 16 MFMAs) per phase.
//     LDS holds 2 K tiles x {X-lo, X-hi, W-lo, W-hi} half-tiles of 128 rows x 128 bytes (128 KiB), filled by
//     global_load_lds_dwordx4 six phases ahead of use: every DMA instruction moves whole 128-byte lines.
//   * the two wave groups (waves 0-3 = frames 0..127 of the tile, waves 4-7 = 128..255; one wave of each per SIMD) run the same
//     program ONE PHASE APART: while one group reads its fragments and issues its DMAs, the other feeds the matrix pipe.
//   * DMA completion is tracked with ONE counted s_waitcnt vmcnt per phase, placed before the raw s_barrier that precedes the
//     first reader; a buffer is restaged two phases after its last reader (details at the loop).
//   * conflict-free LDS: 16-byte chunk c of row rho lives at chunk c ^ ((rho >> 1) & 7); the DMA writes LDS lane-linear, so the
//     XOR is applied to the per-lane SOURCE chunk.
//   * CONV: the X operand is the im2col view of a dilated 1-D convolution — only the per-lane DMA source address changes.
//   * epilogue: (bias is the accumulators' start value) act / BN affine, pack to bf16, swizzled LDS image of the 256 x 256
//     tile, optional per-utterance column sums straight from that image, whole 512-byte rows out in 16-byte stores.
// History of this file (git): a 32-wide K ring with half-line DMAs (-13 %), a v_mfma_f32_32x32x16 variant (-8..12 % on random
// data: lower sustained clock), a persistent variant and a two-phase variant (both +-0) were measured and removed.
#include "common.h"
#include "kernels.h"
#include "gemm_epi.h"

namespace svhip {

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

constexpr int QBM = 256, QBN = 256;
constexpr int HT = 16384;                       // one half-tile: 128 rows x 64 k bf16
constexpr int PW2_LDS = 8 * HT;                 // 128 KiB (also holds the 256 x 256 bf16 output tile)
#ifdef QGROUP_M_OVERRIDE
constexpr int QGROUP_M = QGROUP_M_OVERRIDE;
#else
constexpr int QGROUP_M = 12;
#endif

#ifdef SVHIP_GEMM_DEBUG
constexpr bool DBG2 = true;      // tools/gemm_bench (GemmParams::debug): 1024 / 2048 M-tile groups of 16 / 8, 16384 stage timestamps
#else
constexpr bool DBG2 = false;
#endif
// compile-time ablations (tools/abl_pw2.sh builds one gemm_bench per value; 0 in every shipped build):
//   1 no MFMAs, 2 no operand DMAs, 4 no activation in the epilogue, 8 no output stores, 16 no fragment reads,
//   32 every workgroup DMAs tile (0, 0)'s operands (all L2 hits: separates the memory side of the DMAs from their LDS side)
#ifdef PW2_ABL
constexpr int ABL = PW2_ABL;
#else
constexpr int ABL = 0;
#endif

// H: the 16-bit operand / output type (bf16_t, or f16_t for SVHIP_F16 handles: the same loop, the f16 MFMA opcode and conversions)
template <int EPI, int CONV, typename H = bf16_t>        // CONV: 0 plain, 1 conv-gather, 2 conv-gather + appended pointwise K segment (A3)
__global__ __launch_bounds__(512, 2) void gemm_pw2_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];

    // ---- XCD-aware tile mapping --------------------------------------------------------------
    const int ntm = (p.M + QBM - 1) / QBM, ntn = (p.N + QBN - 1) / QBN;
    const int nwg = ntm * ntn;
    int id = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = id & 7;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
    }
    const int qgm = (DBG2 && (p.debug & 1024)) ? 16 : (DBG2 && (p.debug & 2048)) ? 8 : QGROUP_M;
    const int grp_t = id / (qgm * ntn);
    const int within = id - grp_t * (qgm * ntn);
    const int gm = min(qgm, ntm - grp_t * qgm);
    const int tile_m = grp_t * qgm + within % gm;
    const int tile_n = within / gm;
    const int m0 = tile_m * QBM, n0 = tile_n * QBN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;        // wave group == wm: rows wm*128 .. +127, cols wn*64 .. +63
    const int r16 = lane & 15, q4 = lane >> 4;      // 16x16x32 fragment coordinates
    unsigned long long ts[6] = {0, 0, 0, 0, 0, 0};  // tools/gemm_bench (debug bit 16384): s_memtime at the stage boundaries
#define PW2_STAMP(i) if (DBG2 && (p.debug & 16384) && p.ts) ts[i] = __builtin_readcyclecounter();
    PW2_STAMP(0)

    // accumulators acc16[i][j][e] = channel n0 + wn*64 + j*16 + 4*q4 + e of frame m0 + wm*128 + i*16 + r16; they start at the bias
    f32x4 acc16[8][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f32x4 b0 = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
            const int n = n0 + wn * 64 + j * 16 + 4 * q4;
            if (n < p.N) b0 = *reinterpret_cast<const f32x4*>(p.bias + n);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) acc16[i][j] = b0;
    }

    // ---- four-phase K tiles ---------------------------------------------------------------------------------------
    // Half-tile types: 0 X-lo (frames wm*128 + 0..63 of both wave rows), 1 X-hi (+64..127), 2 W-lo (channels wn*64 + 0..31
    // of the four wave columns), 3 W-hi (+32..63); each 128 rows x 64 k = 16 KiB, two buffers (K-tile parity) per type.
    // Row rho of a half-tile lives at rho*128 + ((chunk ^ ((rho >> 1) & 7)) << 4): conflict-free for the ds_read_b128
    // lane groups of the 16x16x32 fragment pattern, and the DMA (lane-linear in LDS) applies it on the source chunk.
    // Phase g = 4k + ph reads (L) then multiplies (C):  ph 0: X-lo(k) -> quadrant (lo, W-lo)   ph 1: W-hi(k) -> (lo, hi)
    //                                                    ph 2: X-hi(k) -> (hi, W-hi)          ph 3: W-lo(k+1) -> (hi, W-lo(k))
    // so each half-tile is read in exactly one phase, and phase g restages the buffer read in phase g-2 with the
    // half-tile needed in phase g+6 (issue order == need order; W-lo(0) first).  A wave waits (counted vmcnt) in phase g
    // for what phase g+1 reads; the two wave groups run one phase apart.
    const char* src[4][2];
    int dsto[2];
    int cfrm[2][2] = {{0, 0}, {0, 0}}, cchk[2] = {0, 0};   // CONV: frame (within its utterance) and chunk of the X rows
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int rho = (wave * 2 + jj) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((rho >> 1) & 7);
        dsto[jj] = (wave * 2 + jj) * 1024;
        cchk[jj] = c;
#pragma unroll
        for (int ty = 0; ty < 4; ++ty) {
            if (ty < 2) {
                const int m = min(((ABL & 32) ? 0 : m0) + (rho >> 6) * 128 + (ty & 1) * 64 + (rho & 63), p.M - 1);      // (ABL 32: every tile reads tile 0's operands)
                // (CONV: the row start only — the chunk is part of the per-K-tile offset)
                src[ty][jj] = reinterpret_cast<const char*>(p.A) + ((int64_t)m * p.lda + (CONV ? 0 : c * 8)) * 2;
                if (CONV) cfrm[ty][jj] = m - (m / p.T) * p.T;
            } else {
                const int n = min(((ABL & 32) ? 0 : n0) + (rho >> 5) * 64 + (ty & 1) * 32 + (rho & 31), p.Wrows - 1);
                src[ty][jj] = reinterpret_cast<const char*>(p.W) + ((int64_t)n * p.Kp + c * 8) * 2;
            }
        }
    }
    // CONV: the X half-tiles are the im2col view of a dilated 1-D convolution — chunk c of K tile kt is k = kt*64 + 8c ..:
    // tap k / cin of frame t + (tap - taps/2) * dil (reflect or zero padded, chunks past K read the zero page)
    // The address is a 32-bit offset from the lane's own row (a few frames either way), branch-free: a select between two cheap
    // values compiles to v_cndmask, one with a 64-bit multiply in an arm compiles to exec-mask branches between the MFMA phases.
    const float rcin = CONV ? 1.0f / (float)p.cin : 0.0f;
    const bool reflect = p.pad_mode == PAD_REFLECT;
    const int lda2 = p.lda * 2, half = p.taps >> 1;
    const int a3_shift = (CONV == 2 && p.lda3 > 0) ? __builtin_ctz((unsigned)(p.lda / p.lda3)) : 0;
    auto issue = [&](int ty, int kt) {
        char* base = smem + ((kt & 1) * 4 + ty) * HT;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const char* s = src[ty][jj] + (int64_t)kt * 128;
            if (CONV && ty < 2) {
                const int k = kt * 64 + cchk[jj] * 8;
                const int tap = (int)(((float)k + 0.5f) * rcin);          // k / cin (exact: k < 2^16, cin <= 2^10), 3 VALU ops instead of a division
                const int t0 = cfrm[ty][jj];
                const int tt = t0 + (tap - half) * p.dil;
                const bool inside = (unsigned)tt < (unsigned)p.T;
                const int tr = reflect_idx(tt, p.T);                       // == tt when inside
                const bool ok = (k < p.K) & (inside | reflect);
                int offs = (tr - t0) * lda2 + (k - tap * p.cin) * 2;
                asm volatile("" : "+v"(offs));                              // keep the arithmetic out of an `ok` branch
                uint64_t q = reinterpret_cast<uint64_t>(src[ty][jj]) + (int64_t)offs;
                asm volatile("" : "+v"(q));
                s = reinterpret_cast<const char*>(ok ? q : reinterpret_cast<uint64_t>(p.zero_page));
                if (CONV == 2) {
                    // K tiles past the conv columns (uniform per K tile: K % 64 == 0) read row m of A3: same row index as this
                    // lane's X row, so its byte offset is the X row offset scaled by lda3 / lda (a power of two, checked on the host)
                    uint64_t q3 = reinterpret_cast<uint64_t>(p.A3) + ((reinterpret_cast<uint64_t>(src[ty][jj]) - reinterpret_cast<uint64_t>(p.A)) >> a3_shift)
                                  + (uint32_t)((k - p.K) * 2);
                    asm volatile("" : "+v"(q3));
                    s = kt * 64 >= p.K ? reinterpret_cast<const char*>(q3) : s;
                }
            }
            if (!(ABL & 2)) __builtin_amdgcn_global_load_lds((gbl_void*)s, (lds_void*)(base + dsto[jj]), 16, 0, 0);
        }
    };
    auto wait_left = [&](int left) {              // allow `left` half-tiles (2 DMAs each) of this wave to stay in flight
        if (left >= 5) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (left == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (left == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (left == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (left == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    const int nkt = p.Kp / 64;
    const int total = 4 * nkt;                    // half-tiles of this tile's K loop
    issue(2, 0); issue(0, 0); issue(3, 0); issue(1, 0);
    if (nkt > 1) { issue(2, 1); issue(0, 1); issue(3, 1); }
    wait_left(min(7, total) - 2);
    __builtin_amdgcn_s_barrier();                 // W-lo(0), X-lo(0) of every wave have landed
    PW2_STAMP(1)
    if (wm == 1) __builtin_amdgcn_s_barrier();    // group 1 runs one phase behind group 0

    const int xoff = (wm * 64 + r16) * 128, woff = (wn * 32 + r16) * 128;
    const int xkey = ((wm * 64 + r16) >> 1) & 7, wkey = ((wn * 32 + r16) >> 1) & 7;     // rho + 16 i keeps bits 1..3 of rho
    bf16x8 xf[4][2], wlo[2][2], whi[2][2], wnx[2][2];
    {   // W-lo(0)
        const char* b = smem + 2 * HT;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) wlo[j][ks] = *reinterpret_cast<const bf16x8*>(b + woff + j * 2048 + (((ks * 4 + q4) ^ wkey) << 4));
    }
#define PW2_PHASE_END(gidx)                                                                         \
    if (steady) {                                                                                   \
        constexpr int pp_ = ((gidx) + 6) & 3;                                                       \
        const int kk_ = kt + (((gidx) + 6) >> 2);                                                   \
        if (pp_ == 0) issue(0, kk_); else if (pp_ == 1) issue(3, kk_); else if (pp_ == 2) issue(1, kk_); else issue(2, kk_ + 1); \
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");                                           \
        __builtin_amdgcn_s_barrier();                                                               \
    } else {                                                                                        \
        /* the last two K tiles: `left_` phases remain (this one included), so which half-tile is still to be issued and how */ \
        /* many may stay in flight are compile-time constants (as a run-time ladder each phase was a chain of scalar branches) */ \
        constexpr int left_ = 4 * rem - (gidx);                                                     \
        if (left_ > 7) {                                                                            \
            constexpr int pp_ = ((gidx) + 6) & 3;                                                   \
            const int kk_ = kt + (((gidx) + 6) >> 2);                                               \
            if (pp_ == 0) issue(0, kk_); else if (pp_ == 1) issue(3, kk_); else if (pp_ == 2) issue(1, kk_); else issue(2, kk_ + 1); \
        }                                                                                           \
        wait_left((left_ < 8 ? left_ : 8) - 3);                                                     \
        __builtin_amdgcn_s_barrier();                                                               \
    }
#define PW2_MFMA(I0, WARR, J0)                                                                      \
    __builtin_amdgcn_s_setprio(1);                                                                  \
    if (!(ABL & 1))                                                                                 \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                               \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                           \
                acc16[(I0) + i][(J0) + j] = Half16<H>::mfma16(WARR[j][ks], xf[i][ks], acc16[(I0) + i][(J0) + j]); \
    __builtin_amdgcn_s_setprio(0);                                                                  \
    __builtin_amdgcn_s_barrier();
#define PW2_KTILE(steady_, REM_, KT_, WCUR, WNXT)                                                   \
    {                                                                                               \
        constexpr bool steady = (steady_);                                                          \
        constexpr int rem = (REM_);                /* K tiles left, this one included (tail tiles only) */ \
        const int kt = (KT_);                                                                       \
        const char* bb = smem + (kt & 1) * 4 * HT;                                                  \
        /* phase 0: X-lo(kt) x W-lo(kt) */                                                          \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                               \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                        \
                if (!(ABL & 16)) xf[i][ks] = *reinterpret_cast<const bf16x8*>(bb + xoff + i * 2048 + (((ks * 4 + q4) ^ xkey) << 4)); \
        PW2_PHASE_END(0)                                                                            \
        PW2_MFMA(0, WCUR, 0)                                                                        \
        /* phase 1: W-hi(kt); X-lo x W-hi */                                                        \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                               \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                        \
                if (!(ABL & 16)) whi[j][ks] = *reinterpret_cast<const bf16x8*>(bb + 3 * HT + woff + j * 2048 + (((ks * 4 + q4) ^ wkey) << 4)); \
        PW2_PHASE_END(1)                                                                            \
        PW2_MFMA(0, whi, 2)                                                                         \
        /* phase 2: X-hi(kt); X-hi x W-hi */                                                        \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                               \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                        \
                if (!(ABL & 16)) xf[i][ks] = *reinterpret_cast<const bf16x8*>(bb + HT + xoff + i * 2048 + (((ks * 4 + q4) ^ xkey) << 4)); \
        PW2_PHASE_END(2)                                                                            \
        PW2_MFMA(4, whi, 2)                                                                         \
        /* phase 3: W-lo(kt+1) into the other W-lo register set; X-hi x W-lo(kt) */                 \
        if (steady || rem > 1) {                                                                    \
            const char* bn = smem + ((kt + 1) & 1) * 4 * HT + 2 * HT;                               \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                           \
                _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                    \
                    if (!(ABL & 16)) WNXT[j][ks] = *reinterpret_cast<const bf16x8*>(bn + woff + j * 2048 + (((ks * 4 + q4) ^ wkey) << 4)); \
        }                                                                                           \
        PW2_PHASE_END(3)                                                                            \
        PW2_MFMA(4, WCUR, 0)                                                                        \
    }
    int kt0 = 0;
    for (; kt0 + 2 < nkt; ++kt0) {                  // steady state: every issue exists, five half-tiles stay in flight
        PW2_KTILE(true, 0, kt0, wlo, wnx)           // (two K tiles per trip with the W-lo sets swapping roles spills: slower)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) wlo[j][ks] = wnx[j][ks];
    }
    if (nkt >= 2) {                                 // last two K tiles: issues run out, counted drain
        PW2_KTILE(false, 2, kt0, wlo, wnx)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) wlo[j][ks] = wnx[j][ks];
        ++kt0;
    }
    PW2_KTILE(false, 1, kt0, wlo, wnx)
#undef PW2_KTILE
#undef PW2_PHASE_END
#undef PW2_MFMA
    if (wm == 0) __builtin_amdgcn_s_barrier();      // even out the barrier count
    __builtin_amdgcn_s_barrier();                   // every wave is past its last LDS read: the ring becomes the output image
    PW2_STAMP(2)

    // ---- epilogue ------------------------------------------------------------------------------------------------------
    constexpr int ORB = QBN * 2;                    // 512-byte output rows
    // the per-channel constants of every channel group are requested up front (the fragment registers are dead here)
    f32x4 sc4a[4], sh4a[4];
#pragma unroll
    for (int cg = 0; cg < 4; ++cg) {
        const int n = n0 + wn * 64 + cg * 16 + 4 * q4;
        sc4a[cg] = f32x4{1.f, 1.f, 1.f, 1.f}; sh4a[cg] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (n < p.N && p.scale) { sc4a[cg] = *reinterpret_cast<const f32x4*>(p.scale + n); sh4a[cg] = *reinterpret_cast<const f32x4*>(p.shift + n); }
    }
#pragma unroll
    for (int cg = 0; cg < 4; ++cg) {
        const int nl = wn * 64 + cg * 16 + 4 * q4;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ml = wm * 128 + i * 16 + r16;
            float v[4];
            if (ABL & 4) { v[0] = acc16[i][cg][0]; v[1] = acc16[i][cg][1]; v[2] = acc16[i][cg][2]; v[3] = acc16[i][cg][3]; }
            else act4<EPI>(v, acc16[i][cg], sc4a[cg], sh4a[cg]);
            typedef H bf16x4 __attribute__((ext_vector_type(4)));
            bf16x4 o = {static_cast<H>(v[0]), static_cast<H>(v[1]), static_cast<H>(v[2]), static_cast<H>(v[3])};
            *reinterpret_cast<bf16x4*>(smem + ml * ORB + (((nl >> 2) ^ (ml & 15)) << 3)) = o;
        }
    }
    __syncthreads();
    PW2_STAMP(3)
    if (p.colsum) {
        // per-utterance column sums of this tile (feeds the SE mean / the ASP global statistics without
        // another pass over HBM): thread = (column, 128-row half), bf16 values straight from the LDS image
        // thread = (8-byte chunk of 4 channels, 32-row group): 32 ds_read_b64 per thread
        const int c8 = tid & 63, rg = tid >> 6;                 // chunk column 0..63, row group 0..7
        const int rb = (m0 / p.T + 1) * p.T - m0;            // first tile row that belongs to the next utterance
        const int rend = min(256, p.M - m0);
        float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f}, q0[4] = {0.f, 0.f, 0.f, 0.f}, q1[4] = {0.f, 0.f, 0.f, 0.f};
        typedef H bf16x4 __attribute__((ext_vector_type(4)));
        // rows [lo, mid) belong to segment 0, [mid, hi) to segment 1; all three are wave-uniform (rg is the wave index).
        // Almost always a wave's 32 rows sit in one segment: that case runs fully unrolled (32 LDS reads in flight, no selects);
        // a runtime-bounded loop exposes one LDS round trip per row (2.7 us per tile when every wave took it)
        const int lo = rg * 32, hi = min(rg * 32 + 32, rend);
        const int mid = max(lo, min(hi, rb));
        auto add_row = [&](int row, float (&sa)[4], float (&qa)[4]) {
            const bf16x4 v4 = *reinterpret_cast<const bf16x4*>(smem + row * ORB + ((c8 ^ (row & 15)) << 3));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = static_cast<float>(v4[e]);
                sa[e] += x;
                if (p.colsum_sq) qa[e] = fmaf(x, x, qa[e]);
            }
        };
        if (hi - lo == 32 && (mid == hi || mid == lo)) {
            if (mid == hi) {
#pragma unroll
                for (int rr = 0; rr < 32; ++rr) add_row(lo + rr, s0, q0);
            } else {
#pragma unroll
                for (int rr = 0; rr < 32; ++rr) add_row(lo + rr, s1, q1);
            }
        } else {
            for (int row = lo; row < mid; ++row) add_row(row, s0, q0);
            for (int row = mid; row < hi; ++row) add_row(row, s1, q1);
        }
        const int n = n0 + c8 * 4;
        if (n < p.N) {
            const int64_t o = ((int64_t)(tile_m * 8 + rg) * 2) * p.N + n;
            *reinterpret_cast<f32x4*>(p.colsum + o) = f32x4{s0[0], s0[1], s0[2], s0[3]};
            *reinterpret_cast<f32x4*>(p.colsum + o + p.N) = f32x4{s1[0], s1[1], s1[2], s1[3]};
            if (p.colsum_sq) {
                *reinterpret_cast<f32x4*>(p.colsum + p.colsum_stride + o) = f32x4{q0[0], q0[1], q0[2], q0[3]};
                *reinterpret_cast<f32x4*>(p.colsum + p.colsum_stride + o + p.N) = f32x4{q1[0], q1[1], q1[2], q1[3]};
            }
        }
    }
    PW2_STAMP(4)
    char* Yb = reinterpret_cast<char*>(p.Y);
    if (p.R) {
        // residual (RawNet2 conv2 + shortcut): added on the way out, where a lane holds 16 contiguous bytes of a row — the
        // accumulator layout (4 channels of one frame per lane) would read it in scattered 8-byte pieces.  The sum is formed
        // in fp32 from the bf16-rounded tile and rounded once more (<= 1 bf16 ulp against a single rounding).
        const char* Rb = reinterpret_cast<const char*>(p.R);
#pragma unroll
        for (int half = 0; half < 2; ++half) {      // two batches of 8 residual loads in flight per lane
            u32x4 rv[8];
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int idx = (half * 8 + it) * 512 + tid;
                const int m = min(m0 + (idx >> 5), p.M - 1), n = min(n0 + (idx & 31) * 8, p.N - 8);
                rv[it] = *reinterpret_cast<const u32x4*>(Rb + ((int64_t)m * p.ldr + n) * 2);
            }
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int idx = (half * 8 + it) * 512 + tid;
                const int row = idx >> 5, q = idx & 31;
                const int rr = row & 15;
                const u32x4 t = *reinterpret_cast<const u32x4*>(smem + row * ORB + (((2 * q) ^ (rr & 14)) << 3));
                u32x4 d = (rr & 1) ? u32x4{t[2], t[3], t[0], t[1]} : t;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float lo = Half16<H>::lo(d[e]) + Half16<H>::lo(rv[it][e]);
                    const float hi = Half16<H>::hi(d[e]) + Half16<H>::hi(rv[it][e]);
                    d[e] = Half16<H>::pack2(lo, hi);
                }
                const int m = m0 + row, n = n0 + q * 8;
                if (m < p.M && n < p.N) *reinterpret_cast<u32x4*>(Yb + ((int64_t)m * p.ldy + n) * 2) = d;
            }
        }
    } else {
#pragma unroll
        for (int it = 0; it < 16; ++it) {           // 256 rows x 32 16-byte chunks / 512 threads
            const int idx = it * 512 + tid;
            const int row = idx >> 5, q = idx & 31;
            const int rr = row & 15;
            const u32x4 t = *reinterpret_cast<const u32x4*>(smem + row * ORB + (((2 * q) ^ (rr & 14)) << 3));
            const u32x4 d = (rr & 1) ? u32x4{t[2], t[3], t[0], t[1]} : t;
            const int m = m0 + row, n = n0 + q * 8;
            if (m < p.M && n < p.N && !(ABL & 8)) *reinterpret_cast<u32x4*>(Yb + ((int64_t)m * p.ldy + n) * 2) = d;
        }
    }
    if (DBG2 && (p.debug & 16384) && p.ts) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ts[5] = __builtin_readcyclecounter();
        if (tid == 0) {
            unsigned long long* o = reinterpret_cast<unsigned long long*>(p.ts) + (int64_t)blockIdx.x * 8;
            for (int i = 0; i < 6; ++i) o[i] = ts[i];
            unsigned hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            o[6] = hwid; o[7] = xcc;
        }
    }
#undef PW2_STAMP
}

template <int EPI, int CONV, typename H>
hipError_t launch_inst_h(const GemmParams& p, hipStream_t stream) {
    const int ntm = (p.M + QBM - 1) / QBM, ntn = (p.N + QBN - 1) / QBN;
    static DeviceOnce attr;
    if (hipError_t e = set_max_dynamic_lds(attr, reinterpret_cast<const void*>(gemm_pw2_kernel<EPI, CONV, H>), PW2_LDS)) return e;
    hipLaunchKernelGGL((gemm_pw2_kernel<EPI, CONV, H>), dim3(ntm * ntn), dim3(512), PW2_LDS, stream, p);
    return hipGetLastError();
}
// fp16 instances exist for the epilogues RawNet2 uses (ECAPA's 16-bit path is bf16): none / LeakyReLU forms
template <int EPI, int CONV>
hipError_t launch_inst(const GemmParams& p, hipStream_t stream) {
    if (p.f16) {
        if constexpr (EPI == EPI_GELU || EPI == EPI_RELU) return hipErrorInvalidValue;
        else return launch_inst_h<EPI, CONV, f16_t>(p, stream);
    }
    return launch_inst_h<EPI, CONV, bf16_t>(p, stream);
}

}  // namespace

bool gemm_pw2_supported(const GemmParams& p, bool bf16) {
    if (!bf16 || p.out_f32 || p.bias_utt || p.A2) return false;
    if (p.R && (p.colsum || p.ldr % 8 != 0 || (reinterpret_cast<uintptr_t>(p.R) & 15))) return false;      // (the column sums are taken before the residual)
    if (p.act2 != ACT_NONE && !(p.act2 == ACT_LRELU03 && p.act1 == ACT_NONE)) return false;
    if (p.colsum && (p.T < 256 || p.M % p.T != 0)) return false;      // at most one utterance boundary per 256-row tile
    if (!(p.act1 == ACT_NONE || p.act1 == ACT_RELU || p.act1 == ACT_GELU || p.act1 == ACT_LRELU03)) return false;
    if (p.f16 && (p.act1 == ACT_RELU || p.act1 == ACT_GELU)) return false;          // (no fp16 instances of ECAPA's epilogues)
    if (p.N < 256 || p.Kp % 64 != 0 || p.N % 8 != 0 || p.lda % 8 != 0 || p.ldy % 8 != 0) return false;
    if ((reinterpret_cast<uintptr_t>(p.A) | reinterpret_cast<uintptr_t>(p.W) | reinterpret_cast<uintptr_t>(p.Y)) & 15) return false;
    if (p.bias && (reinterpret_cast<uintptr_t>(p.bias) & 15)) return false;
    if (p.scale && ((reinterpret_cast<uintptr_t>(p.scale) | reinterpret_cast<uintptr_t>(p.shift)) & 15)) return false;
    if (p.A3) {              // appended pointwise segment: conv-gather, no activation, tile-aligned segments, lda = lda3 << s
        if (p.taps <= 1 || p.act1 != ACT_NONE || p.act2 != ACT_NONE || p.K % 64 != 0 || p.K3 <= 0 || p.K3 % 64 != 0 || p.Kp != p.K + p.K3) return false;
        if (p.lda3 <= 0 || p.lda % p.lda3 != 0 || ((p.lda / p.lda3) & (p.lda / p.lda3 - 1)) != 0 || p.lda3 % 8 != 0 || p.K3 > p.lda3) return false;
        if (reinterpret_cast<uintptr_t>(p.A3) & 15) return false;
    }
    if (p.taps > 1) {        // conv-gather: 16-byte chunks must not straddle taps; needs the zero page for padded k / frames
        if (!p.zero_page || p.cin % 8 != 0 || p.taps * p.cin != p.K || p.T <= 0 || p.M % p.T != 0) return false;
        if (p.pad_mode == PAD_REFLECT && (p.taps / 2) * p.dil >= p.T) return false;
        if (p.Kp >= 65536 || p.cin > 1024) return false;       // the kernel's k / cin is a float multiply
        return true;
    }
    return p.K == p.Kp;
}

hipError_t launch_gemm_pw2(const GemmParams& p, hipStream_t stream) {
    if (!gemm_pw2_supported(p, true) || p.M <= 0 || p.Wrows < p.N) return hipErrorInvalidValue;
    const bool conv = p.taps > 1;
    if (p.A3) return launch_inst<EPI_NONE, 2>(p, stream);
    if (p.act2 == ACT_LRELU03) return conv ? launch_inst<EPI_BN_LRELU03, 1>(p, stream) : launch_inst<EPI_BN_LRELU03, 0>(p, stream);
    switch (p.act1) {
        case ACT_NONE: return conv ? launch_inst<EPI_NONE, 1>(p, stream) : launch_inst<EPI_NONE, 0>(p, stream);
        case ACT_RELU: return conv ? launch_inst<EPI_RELU, 1>(p, stream) : launch_inst<EPI_RELU, 0>(p, stream);
        case ACT_GELU: return conv ? launch_inst<EPI_GELU, 1>(p, stream) : launch_inst<EPI_GELU, 0>(p, stream);
        case ACT_LRELU03: return conv ? launch_inst<EPI_LRELU03, 1>(p, stream) : launch_inst<EPI_LRELU03, 0>(p, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace svhip
