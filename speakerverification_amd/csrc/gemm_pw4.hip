// gemm_pw4.hip — PERSISTENT bf16 pointwise GEMM, 256 x 256 tile on FOUR waves (128 x 128 per wave, one wave per SIMD), gfx950.
//
// Round 6.  Same contract as gemm_pw3.hip (Y = epi(A . W^T), 16-bit in / fp32 accumulate / 16-bit out, bias in the accumulators'
// start value, activation + BN affine + per-utterance column sums from the accumulators, tiles walked by persistent workgroups in
// per-XCD bands).  What differs is the cut of the tile over waves: the vendor's plain GEMM of ECAPA's two pointwise shapes runs
// 13 - 16 % faster than gemm_pw3 without its epilogue (profiles/r06_pw3_yardstick.txt: hipBLASLt picks a 256 x 256 x 64 macro tile
// on four waves with 128 x 128 wave tiles and direct-to-LDS loads), and a 128 x 128 wave tile reads 2 / 3 of the LDS fragment bytes
// per MFMA of gemm_pw3's 128 x 64 one (16 + 16 fragment reads per 128 MFMAs instead of 16 + 8 per 64).
//
// Structure of a K tile (64 k; two k steps of 32), per wave:
//   fragments of k step 1 are read while the 64 MFMAs of k step 0 run, then ONE barrier per K tile (every wave is done reading the
//   K tile's buffer and its own DMAs of the next K tile have landed), the DMAs of the K tile after next go out into the buffer just
//   freed, fragments of the next K tile's k step 0 are read while the 64 MFMAs of k step 1 run.  128 KiB ring = two K tiles of
//   [X 256 x 64 | W 256 x 64]; LDS rows, swizzle and the DMA source addressing are gemm_pw3's.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "kernels.h"
#include "gemm_epi.h"

namespace svhip {

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

constexpr int HT4 = 16384;                      // one half-tile: 128 rows x 64 k of a 16-bit type
constexpr int RING4 = 8 * HT4;                  // two K tiles of four half-tiles
constexpr int CST4 = 3072;                      // bias | scale | shift of an N-tile, 256 floats each
constexpr int PW4_LDS = RING4 + 2 * CST4;
constexpr int PGROUP4_M = 8;
// compile-time ablations (tools only; 0 in every shipped build): 1 no operand DMAs, 2 no K-loop barrier (results are then wrong)
#ifndef PW4_ABL
#define PW4_ABL 0
#endif
#ifndef PW4_SKEW
#define PW4_SKEW 0
#endif

template <int EPI, int CS, typename H>
__global__ __launch_bounds__(256, 1) void gemm_pw4_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ntm = (p.M + 255) / 256, ntn = (p.N + 255) / 256;
    const int ntiles = ntm * ntn;
    const int G = gridDim.x;
    int perm = blockIdx.x;
    {
        const int q = G >> 3, r = G & 7, xcd = perm & 7;
        perm = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (perm >> 3);
    }
    auto tile_of = [&](int w, int& tm, int& tn) {
        const int per = PGROUP4_M * ntn;
        const int grp = w / per;
        const int within = w - grp * per;
        const int gm = min(PGROUP4_M, ntm - grp * PGROUP4_M);
        const int tnn = within / gm;
        tm = grp * PGROUP4_M + (within - tnn * gm);
        tn = tnn;
    };
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    auto lane_now = [&]() { int t = tid; asm volatile("" : "+v"(t)); return t & 63; };

    // ---- operand DMA: a half-tile is 16 pieces of 8 rows x 128 bytes; this wave moves pieces 4 wave .. 4 wave + 3 of each --------
    uint32_t xo[2][4], wo[2][4];
    const char* abase = nullptr;
    const char* wbase = nullptr;
    auto set_src = [&](int m0, int n0) {
        const int lane = lane_now();
        abase = reinterpret_cast<const char*>(p.A) + (int64_t)m0 * p.lda * 2;
        wbase = reinterpret_cast<const char*>(p.W) + (int64_t)n0 * p.Kp * 2;
        const int mmax = p.M - 1 - m0, nmax = p.Wrows - 1 - n0;
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) {
            const int rho = (wave * 4 + pc) * 8 + (lane >> 3);          // row inside the half-tile
            const int c = (lane & 7) ^ ((rho >> 1) & 7);
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int m = min(hf * 128 + rho, mmax);
                xo[hf][pc] = __umul24((uint32_t)m, (uint32_t)(p.lda * 2)) + (uint32_t)(c * 16);
                const int n = min(hf * 128 + rho, nmax);
                wo[hf][pc] = __umul24((uint32_t)n, (uint32_t)(p.Kp * 2)) + (uint32_t)(c * 16);
            }
        }
    };
    u32x4 dW[8];                                    // (PW4_ABL & 4 only)
    // all four half-tiles of K tile kt of the current source -> ring buffer `buf`
    auto issue_ktile = [&](int kt, int buf) {
        char* base = smem + buf * 4 * HT4 + wave * 4096;
        const char* ua = abase + (int64_t)kt * 128;
        const char* uw = wbase + (int64_t)kt * 128;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int pc = 0; pc < 4; ++pc) {
                uint32_t o = wo[hf][pc];
                asm("" : "+v"(o));           // (not volatile: opaque to the optimiser — SGPR base + 32-bit VGPR offset addressing — but no scheduling boundary)
                if (PW4_ABL & 4) {             // ablation: the W pieces as plain loads into registers nobody reads (results wrong; what an LDS-DMA costs beside a load)
                    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dW[hf * 4 + pc]) : "v"(o), "s"(uw) : "memory");      // (dW stays live until the K tile's vmcnt wait)
                } else
                if (!(PW4_ABL & 1)) __builtin_amdgcn_global_load_lds((gbl_void*)(uw + o), (lds_void*)(base + (2 + hf) * HT4 + pc * 1024), 16, 0, 0);
            }
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int pc = 0; pc < 4; ++pc) {
                uint32_t o = xo[hf][pc];
                asm("" : "+v"(o));
                if (!(PW4_ABL & 1)) __builtin_amdgcn_global_load_lds((gbl_void*)(ua + o), (lds_void*)(base + hf * HT4 + pc * 1024), 16, 0, 0);
            }
    };
    auto issue_consts = [&](int n0, int par) {
        if (wave < 3) {
            const float* s = (wave == 0 ? p.bias : wave == 1 ? p.scale : p.shift) + min(n0 + lane_now() * 4, p.N - 4);
            __builtin_amdgcn_global_load_lds((gbl_void*)s, (lds_void*)(smem + RING4 + par * CST4 + wave * 1024), 16, 0, 0);
        }
    };

    const int nkt = p.Kp * 2 / 128;               // K tiles per row: even, >= 2 (host check)

    int w = perm, tm, tn;
    tile_of(w, tm, tn);
    set_src(tm * 256, tn * 256);
    issue_consts(tn * 256, 0);
    issue_ktile(0, 0);
    issue_ktile(1, 1);

    // fragment addressing: lane (r16, q4) reads row 16 i + r16, 16-byte chunk (4 ks + q4) ^ key
    bf16x8 xa[8], wa[8], xb[8], wb[8];
    auto read_frags = [&](bf16x8 (&xf)[8], bf16x8 (&wf)[8], int buf, int ks) {
        const int lane = lane_now();
        const int r16 = lane & 15, q4 = lane >> 4;
        const int key = (r16 >> 1) & 7;
        const char* bx = smem + buf * 4 * HT4 + wm * HT4 + r16 * 128 + (((ks * 4 + q4) ^ key) << 4);
        const char* bw = smem + buf * 4 * HT4 + (2 + wn) * HT4 + r16 * 128 + (((ks * 4 + q4) ^ key) << 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(bw + j * 2048);
#pragma unroll
        for (int i = 0; i < 8; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(bx + i * 2048);
    };

    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");          // the constants and K tile 0 of every wave ...
    __builtin_amdgcn_s_barrier();                              // ... have landed
    asm volatile("" ::: "memory");
    read_frags(xa, wa, 0, 0);
    bool stores_behind = false;                                // the wave's queue holds the previous tile's epilogue stores behind its DMAs

    for (int it = 0;; ++it) {
        const int m0 = tm * 256, n0 = tn * 256, par = it & 1;
        const char* cb = smem + RING4 + par * CST4;
        const int w_next = w + G;
        const bool more = w_next < ntiles;
        // (a workgroup's last tile fetches its own first two K tiles once more instead of a next tile's: the loop body has no
        //  conditional issue; the DMAs are drained before the wave ends)
        int tm_n = tm, tn_n = tn;
        if (more) tile_of(w_next, tm_n, tn_n);

        // the accumulators start at zero (the first MFMA of each takes the constant as its C operand); the bias joins in the epilogue
        f32x4 acc[8][8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#define PW4_MFMA(XF, WF, I0, I1)                                                                    \
    _Pragma("unroll") for (int i = (I0); i < (I1); ++i)                                             \
        _Pragma("unroll") for (int j = 0; j < 8; ++j)                                               \
            acc[i][j] = Half16<H>::mfma16(WF[j], XF[i], acc[i][j]);
#ifndef PW4_VARIANT
#define PW4_VARIANT 3
#endif
        // One K tile.  MODE 0: steady state; 1: the tile's first K tile (the previous tile's stores may sit behind the awaited DMAs);
        // 2: K tile nkt - 2 (the DMA source moves on to the next tile).  SRC_KT: the K tile of the current source the issue fetches.
        // Schedule (PW4_VARIANT 3): the reads of k step 1 go out under the first 32 MFMAs of k step 0, the barrier sits at the quarter
        // point, the 16 operand DMAs follow four MFMAs apart (back to back they queue on each other; an LDS read may not pass a DMA —
        // both touch the LDS — so the next K tile's first reads come last, two MFMAs apart, under the tail of k step 1).
#define PW4_WAIT_BARRIER(MODE)                                                                      \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     /* this wave is done reading `buf` */ \
        /* its DMAs of the next K tile (issued a K tile ago) have landed */                         \
        if ((MODE) == 1 && stores_behind) {     /* 32 output stores + 16 per kind of column sum (CS = 2: one of the 64 must be back) */ \
            if (CS == 0) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");                          \
            else if (CS == 1) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");                     \
            else asm volatile("s_waitcnt vmcnt(63)" ::: "memory");                                  \
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                     \
        if (PW4_ABL & 4) asm volatile("" :: "v"(dW[0]), "v"(dW[1]), "v"(dW[2]), "v"(dW[3]), "v"(dW[4]), "v"(dW[5]), "v"(dW[6]), "v"(dW[7])); \
        if (!(PW4_ABL & 2)) __builtin_amdgcn_s_barrier();                                           \
        /* skew: wave w runs PW4_SKEW cycles behind wave w - 1 until the next barrier, so that the four waves' operand DMAs do not */ \
        /* reach the CU's one address path in the same cycle */                                     \
        if (PW4_SKEW > 0) {                                                                         \
            if (wave & 1) { _Pragma("unroll") for (int z = 0; z < PW4_SKEW / 16; ++z) asm volatile("s_nop 15"); }      \
            if (wave & 2) { _Pragma("unroll") for (int z = 0; z < PW4_SKEW / 8; ++z) asm volatile("s_nop 15"); }       \
        }                                                                                           \
        asm volatile("" ::: "memory");                                                              \
        __builtin_amdgcn_sched_barrier(0);
#if PW4_VARIANT == 3
#define PW4_KTILE(MODE, KT, SRC_KT)                                                                 \
    {                                                                                               \
        const int buf = (KT) & 1;                                                                   \
        read_frags(xb, wb, buf, 1);                                                                 \
        PW4_MFMA(xa, wa, 0, 4)                                                                      \
        _Pragma("unroll") for (int g = 0; g < 16; ++g) {                                            \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                      \
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                      \
        }                                                                                           \
        PW4_WAIT_BARRIER(MODE)                                                                      \
        if ((MODE) == 2) { set_src(tm_n * 256, tn_n * 256); issue_consts(tn_n * 256, par ^ 1); }    \
        issue_ktile((SRC_KT), buf);                                                                 \
        PW4_MFMA(xa, wa, 4, 8)                                                                      \
        read_frags(xa, wa, buf ^ 1, 0);                                                             \
        PW4_MFMA(xb, wb, 0, 8)                                                                      \
        _Pragma("unroll") for (int g = 0; g < 16; ++g) {                                            \
            __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);                                      \
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                      \
        }                                                                                           \
        _Pragma("unroll") for (int g = 0; g < 16; ++g) {                                            \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                      \
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                      \
        }                                                                                           \
        asm volatile("" ::: "memory");                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                          \
    }
#else
#define PW4_KTILE(MODE, KT, SRC_KT)                                                                 \
    {                                                                                               \
        const int buf = (KT) & 1;                                                                   \
        /* region 1: k step 0's MFMAs over the reads of k step 1 */                                 \
        read_frags(xb, wb, buf, 1);                                                                 \
        PW4_MFMA(xa, wa, 0, 8)                                                                      \
        _Pragma("unroll") for (int g = 0; g < 16; ++g) {                                            \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                      \
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                      \
        }                                                                                           \
        PW4_WAIT_BARRIER(MODE)                                                                      \
        /* region 2: the K tile after next into the buffer just freed; k step 1's MFMAs over the next K tile's first reads */ \
        if ((MODE) == 2) { set_src(tm_n * 256, tn_n * 256); issue_consts(tn_n * 256, par ^ 1); }    \
        issue_ktile((SRC_KT), buf);                                                                 \
        read_frags(xa, wa, buf ^ 1, 0);                                                             \
        PW4_MFMA(xb, wb, 0, 8)                                                                      \
        _Pragma("unroll") for (int g = 0; g < 16; ++g) {                                            \
            __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);                                      \
            __builtin_amdgcn_sched_group_barrier(0x008, PW4_VARIANT == 1 ? 3 : 2, 0);               \
        }                                                                                           \
        _Pragma("unroll") for (int g = 0; g < 16; ++g) {                                            \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                      \
            __builtin_amdgcn_sched_group_barrier(0x008, PW4_VARIANT == 1 ? 1 : 2, 0);               \
        }                                                                                           \
        asm volatile("" ::: "memory");                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                          \
    }
#endif
        PW4_KTILE(1, 0, 2)
        for (int kt = 1; kt + 2 < nkt; ++kt) PW4_KTILE(0, kt, kt + 2)
        PW4_KTILE(2, nkt - 2, 0)
        PW4_KTILE(0, nkt - 1, 1)
#undef PW4_KTILE
#undef PW4_WAIT_BARRIER
#undef PW4_MFMA
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);

        // ---- epilogue from the accumulators: lane (r16, q4) holds channels n0 + wn*128 + j*16 + 4 q4 + e of frame m0 + wm*128 + i*16 + r16
        {
            const int lane_e = lane_now();
            const int r16e = lane_e & 15, q4e = lane_e >> 4;
            char* Yb = reinterpret_cast<char*>(p.Y);
#pragma unroll
            for (int jp = 0; jp < 4; ++jp) {
                f32x4 bi[2], sc[2], sh[2];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int nl = wn * 128 + (2 * jp + jj) * 16 + 4 * q4e;
                    bi[jj] = *reinterpret_cast<const f32x4*>(cb + nl * 4);
                    sc[jj] = *reinterpret_cast<const f32x4*>(cb + 1024 + nl * 4);
                    sh[jj] = *reinterpret_cast<const f32x4*>(cb + 2048 + nl * 4);
                }
                const int c0 = n0 + wn * 128 + jp * 32 + (q4e & 1) * 16 + (q4e >> 1) * 8;
                char* yl = Yb + ((int64_t)(m0 + wm * 128 + r16e) * p.ldy + c0) * 2;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float v0[4], v1[4];
                    act4<EPI>(v0, acc[i][2 * jp] + bi[0], sc[0], sh[0]);
                    act4<EPI>(v1, acc[i][2 * jp + 1] + bi[1], sc[1], sh[1]);
                    if (CS) {
                        acc[i][2 * jp] = f32x4{v0[0], v0[1], v0[2], v0[3]};
                        acc[i][2 * jp + 1] = f32x4{v1[0], v1[1], v1[2], v1[3]};
                    }
                    const auto s0 = __builtin_amdgcn_permlane16_swap(Half16<H>::pack2(v0[0], v0[1]), Half16<H>::pack2(v1[0], v1[1]), false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(Half16<H>::pack2(v0[2], v0[3]), Half16<H>::pack2(v1[2], v1[3]), false, false);
                    const int m = m0 + wm * 128 + i * 16 + r16e;
                    if (m < p.M) *reinterpret_cast<u32x4*>(yl + (int64_t)i * 16 * p.ldy * 2) = u32x4{s0[0], s1[0], s0[1], s1[1]};
                }
            }
            if (CS) {
                // per-utterance column sums of this wave's 128 frames x 128 channels (gemm_pw3's layout: 2 row groups of 128 rows per tile):
                //   colsum[((tm*2 + wm)*2 + seg) * N + n], seg 0 = the utterance of the tile's first row, seg 1 = the next one
                const int lo = wm * 128;
                const int rb = (m0 / p.T + 1) * p.T - m0;         // first tile row that belongs to the next utterance
                const int rend = min(256, p.M - m0);
                const bool whole = (lo + 128 <= rend) && (lo + 128 <= rb || lo >= rb);      // wave-uniform: one segment, every row valid
                float* csp = p.colsum + ((int64_t)(tm * 2 + wm) * 2) * p.N + n0 + wn * 128 + 4 * q4e;
                const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
                if (whole) {
                    const int sg = lo >= rb ? 1 : 0;
#pragma unroll
                    for (int kind = 0; kind < CS; ++kind) {
                        float* cs = csp + (kind ? p.colsum_stride : 0);
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            f32x4 sm = zero4;
#pragma unroll
                            for (int i = 0; i < 8; ++i) sm += kind ? acc[i][j] * acc[i][j] : acc[i][j];
#pragma unroll
                            for (int e = 0; e < 4; ++e) sm[e] = row16_sum(sm[e]);
                            if (r16e == 0) {
                                *reinterpret_cast<f32x4*>(cs + j * 16 + (int64_t)sg * p.N) = sm;
                                *reinterpret_cast<f32x4*>(cs + j * 16 + (int64_t)(1 - sg) * p.N) = zero4;
                            }
                        }
                    }
                } else {
                    // an utterance boundary inside this wave's rows, or rows past M: every row weighted on its own
#pragma unroll
                    for (int kind = 0; kind < CS; ++kind) {
                        float* cs = csp + (kind ? p.colsum_stride : 0);
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            f32x4 s0 = zero4, s1 = zero4;
#pragma unroll
                            for (int i = 0; i < 8; ++i) {
                                const int row = lo + i * 16 + r16e;
                                const float w0 = (row < rb && row < rend) ? 1.0f : 0.0f;
                                const float w1 = (row >= rb && row < rend) ? 1.0f : 0.0f;
                                const f32x4 v = kind ? acc[i][j] * acc[i][j] : acc[i][j];
                                s0 += v * w0; s1 += v * w1;
                            }
#pragma unroll
                            for (int e = 0; e < 4; ++e) { s0[e] = row16_sum(s0[e]); s1[e] = row16_sum(s1[e]); }
                            if (r16e == 0) {
                                *reinterpret_cast<f32x4*>(cs + j * 16) = s0;
                                *reinterpret_cast<f32x4*>(cs + j * 16 + p.N) = s1;
                            }
                        }
                    }
                }
            }
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (!more) break;
        stores_behind = m0 + 256 <= p.M;       // (a tile with masked rows: some waves issued fewer stores — the strict count is used)
        w = w_next; tm = tm_n; tn = tn_n;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the last tile's two surplus K tiles
}

template <int EPI, int CS, typename H>
hipError_t launch_inst4(const GemmParams& p, hipStream_t stream) {
    const int ntiles = ((p.M + 255) / 256) * (p.N / 256);
    static DeviceOnce attr;
    if (hipError_t e = set_max_dynamic_lds(attr, reinterpret_cast<const void*>(gemm_pw4_kernel<EPI, CS, H>), PW4_LDS)) return e;
    const int grid = min(pw3_grid_cap(p), ntiles);
    hipLaunchKernelGGL((gemm_pw4_kernel<EPI, CS, H>), dim3(grid), dim3(256), PW4_LDS, stream, p);
    return hipGetLastError();
}

}  // namespace

// the shapes gemm_pw3's plain pointwise form takes (bf16 or fp16 operands, N % 256 == 0, an even count of 64-wide K tiles, no residual)
bool gemm_pw4_supported(const GemmParams& p, bool bf16) {
    if (!gemm_pw3_supported(p, bf16) || p.f16 || p.Kp < 256) return false;
    const int epi_ok = p.act1 == ACT_GELU || ((p.act1 == ACT_NONE || p.act1 == ACT_RELU) && !p.colsum);
    const int ntiles = ((p.M + 255) / 256) * (p.N / 256);
    // whole tiles only: a grid that does not fill the chip stays on gemm_pw3 (its column halves put every CU to work)
    return epi_ok && ntiles > pw3_grid_cap(p);
}

hipError_t launch_gemm_pw4(const GemmParams& p, hipStream_t stream) {
    const int cs = p.colsum ? (p.colsum_sq ? 2 : 1) : 0;
    const int epi = p.act1 == ACT_GELU ? EPI_GELU : p.act1 == ACT_RELU ? EPI_RELU : EPI_NONE;
    if (p.act2 != ACT_NONE || p.f16) return hipErrorInvalidValue;
#define PW4_CASE(E, C) if (epi == E && cs == C) return launch_inst4<E, C, bf16_t>(p, stream);
    PW4_CASE(EPI_GELU, 0) PW4_CASE(EPI_GELU, 1) PW4_CASE(EPI_GELU, 2)
    PW4_CASE(EPI_NONE, 0) PW4_CASE(EPI_RELU, 0)
#undef PW4_CASE
    return hipErrorInvalidValue;
}

}  // namespace svhip
