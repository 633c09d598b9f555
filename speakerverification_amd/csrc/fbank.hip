// fbank.hip — pre-emphasis + STFT power spectrum + Slaney mel bank on gfx950.
//
// Replaces Sequential(PreEmphasis, nnAudio MelSpectrogram) (reference
// models/FeatureExtraction/feature.py:66-94, src/utils.py:53-71): wav (B, L) -> mel power (B, n_mels, T).
//
// One workgroup = 32 consecutive frames of one utterance, 3 waves.
//   1. the pre-emphasised, reflect-padded samples the 32 frames touch (31*hop + win_length) are built
//      once in LDS (coalesced HBM read of the waveform tile);
//   2. the windowed DFT is a (32 frames x win_length taps) x (taps x 2*n_bins) product on the exact
//      fp32 MFMA (v_mfma_f32_32x32x2_f32): the A fragments are ds_read_b128 of 4 consecutive taps
//      straight out of the sample buffer (frames overlap, so no im2col copy exists anywhere), the B
//      fragments (windowed cos / sin, laid out per lane as float4) stream from L2.  re and im of a bin
//      land in the same lane, so |X|^2 is formed in registers;
//   3. the 32 x n_bins power tile goes through LDS (stride 289: conflict-free both ways) and the
//      sparse triangular mel filters are applied from there; frames are written coalesced along T.
// Only the non-zero window taps (200 of 512) are multiplied.
// Pipe counters of the bf16x3 form at B = 256 (round 3, profiles/r03_pipe_counters.json; 190 - 230 us by box): matrix pipe busy
// 0.17 of the CU cycles, LDS array 6.5 %, 47 M vector instructions per launch against 2.3 M MFMAs (4.7 k per wave: sample staging
// with its reflect arithmetic, basis addressing, a correctly rounded software sqrt per bin, the sparse mel sums), and the waves
// spend 49 % of their cycles in s_waitcnt / barriers: three-wave workgroups of 10 us each, 2.25 waves per SIMD, every k step
// a round trip to L2 for 12 KB of basis per wave.  Two cuts of the vector work alone (|X|^2 as re^2 + im^2 on the split path,
// one running basis offset per k step) measured 205 against 190 us on one box (tools/fbank_bench.py) and were not kept: the
// kernel is latency-bound, not issue-bound.  The structural fix (9-wave workgroups of 128 frames so that a basis fragment feeds
// 12 MFMAs instead of 3) is the open lead.
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace svhip {

namespace {

constexpr int FB_FRAMES = 32;
constexpr int FB_THREADS = 192;
constexpr int FB_PT_STRIDE = 289;     // 9*32 + 1
constexpr int FB_PAIRS_PER_WAVE = 3;

// SPLIT = false: exact fp32 MFMA (v_mfma_f32_32x32x2_f32), the 1e-4-parity path.
// SPLIT = true : bf16x3 — samples and basis are split into bf16 hi + lo parts and the product is
//                hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 (dropped term ~2^-16 relative): 5x
//                fewer matrix-pipe cycles, used by bf16-compute handles whose features are rounded to
//                bf16 (2^-8) right after.
template <bool SPLIT>
__global__ __launch_bounds__(FB_THREADS) void fbank_kernel(FbankTables tb, const float* __restrict__ wav,
                                                           int L, int T, float* __restrict__ mel) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ns = (FB_FRAMES - 1) * tb.hop + tb.win_length;       // samples touched by the tile
    const int ns_pad = ((ns + 15) & ~15) + 16;                     // + one zero-basis k-step of slack for SPLIT
    float* ys = reinterpret_cast<float*>(smem);                    // SPLIT: [ns_pad] bf16 hi | [ns_pad] bf16 lo
    float* pt = ys + ns_pad;                                       // [32][289]
    float* melw = pt + FB_FRAMES * FB_PT_STRIDE;                   // packed non-zero mel weights (LDS copy)

    const int b = blockIdx.y;
    const int f0 = blockIdx.x * FB_FRAMES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* __restrict__ x = wav + (int64_t)b * L;

    for (int i = tid; i < tb.n_melw; i += FB_THREADS) melw[i] = tb.mel_w[i];
    // ---- 1. pre-emphasised + reflect-padded samples -> LDS -------------------------------------
    const int j0 = f0 * tb.hop + tb.lpad - tb.n_fft / 2;
    const float coef = tb.preemph;
    // four samples per thread in flight per trip (one per trip exposed an HBM round trip per trip, ~15 trips per tile)
    for (int i0 = tid; i0 < ns_pad; i0 += 4 * FB_THREADS) {
        float v[4], prev[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * FB_THREADS;
            int jj = j0 + i;
            jj = jj < 0 ? -jj : jj;
            jj = jj >= L ? 2 * (L - 1) - jj : jj;
            jj = max(0, min(jj, L - 1));
            v[u] = i < ns_pad ? x[jj] : 0.0f;
            prev[u] = (i < ns_pad && coef >= 0.0f) ? x[jj == 0 ? 1 : jj - 1] : 0.0f;   // F.pad(reflect,(1,0)): x[-1] := x[1]
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * FB_THREADS;
            if (i >= ns_pad) break;
            float w = v[u];
            if (coef >= 0.0f) w = __fadd_rn(__fmul_rn(-coef, prev[u]), w);     // conv1d with taps [-coef, 1]
            if (SPLIT) {
                const bf16_t hi = static_cast<bf16_t>(w);
                reinterpret_cast<bf16_t*>(ys)[i] = hi;
                reinterpret_cast<bf16_t*>(ys)[ns_pad + i] = static_cast<bf16_t>(w - static_cast<float>(hi));
            } else {
                ys[i] = w;
            }
        }
    }
    __syncthreads();

    // ---- 2. windowed DFT on the fp32 MFMA --------------------------------------------------------
    const int r = lane & 31, h = lane >> 5;
    f32x16 acc[FB_PAIRS_PER_WAVE][2];
#pragma unroll
    for (int a = 0; a < FB_PAIRS_PER_WAVE; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][c][e] = 0.0f;

    if (SPLIT) {
        const bf16x8* __restrict__ bh = reinterpret_cast<const bf16x8*>(tb.basis_hi);
        const bf16x8* __restrict__ bl = reinterpret_cast<const bf16x8*>(tb.basis_lo);
        const bf16_t* yh = reinterpret_cast<const bf16_t*>(ys) + r * tb.hop + 8 * h;
        const bf16_t* yl = yh + ns_pad;
        // (a register double-buffer of the basis fragments was tried: 212 VGPRs cost a wave per SIMD and
        //  lost more than the prefetch gained; occupancy hides the L2 latency better here)
        for (int kk = 0; kk < tb.n_k16; ++kk) {
            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(yh + 16 * kk);
            const bf16x8 al = *reinterpret_cast<const bf16x8*>(yl + 16 * kk);
#pragma unroll
            for (int a = 0; a < FB_PAIRS_PER_WAVE; ++a) {
                const int pair = wave * FB_PAIRS_PER_WAVE + a;
                if (pair < tb.n_pairs) {
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const int64_t bi = ((int64_t)(kk * tb.n_pairs + pair) * 2 + c) * 64 + lane;
                        const bf16x8 b_hi = bh[bi], b_lo = bl[bi];
                        acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b_hi, acc[a][c], 0, 0, 0);   // small terms first
                        acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b_lo, acc[a][c], 0, 0, 0);
                        acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b_hi, acc[a][c], 0, 0, 0);
                    }
                }
            }
        }
    } else {
        const f32x4* __restrict__ basis = reinterpret_cast<const f32x4*>(tb.basis);
        const float* arow = ys + r * tb.hop + 4 * h;
        for (int q = 0; q < tb.n_q; ++q) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(arow + 8 * q);
#pragma unroll
            for (int a = 0; a < FB_PAIRS_PER_WAVE; ++a) {
                const int pair = wave * FB_PAIRS_PER_WAVE + a;
                if (pair < tb.n_pairs) {
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const f32x4 b4 = basis[((int64_t)(q * tb.n_pairs + pair) * 2 + c) * 64 + lane];
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j], b4[j], acc[a][c], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- 3. power -> LDS (frame-major), then sparse mel filters ----------------------------------
#pragma unroll
    for (int a = 0; a < FB_PAIRS_PER_WAVE; ++a) {
        const int pair = wave * FB_PAIRS_PER_WAVE + a;
        if (pair < tb.n_pairs) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float re = acc[a][0][e], im = acc[a][1][e];
                const float mag = __fsqrt_rn(__fadd_rn(__fmul_rn(re, re), __fmul_rn(im, im)));   // 'Magnitude'
                const int i = (e & 3) + 8 * (e >> 2) + 4 * h;
                pt[i * FB_PT_STRIDE + pair * 32 + r] = __fmul_rn(mag, mag);                      // ** 2.0
            }
        }
    }
    __syncthreads();

    for (int idx = tid; idx < FB_FRAMES * tb.n_mels; idx += FB_THREADS) {
        const int i = idx & 31, m = idx >> 5;
        const int f = f0 + i;
        const int st = tb.mel_start[m], ln = tb.mel_len[m];
        const float* w = melw + tb.mel_off[m];
        const float* prow = pt + i * FB_PT_STRIDE + st;
        float s = 0.0f;
        for (int k = 0; k < ln; ++k) s = fmaf(w[k], prow[k], s);
        if (f < T) mel[((int64_t)b * tb.n_mels + m) * T + f] = s;
    }
}

// ---- the bf16x3 form, round 3: 64 frames per workgroup, one re/im pair of 32 bins per wave -----------------------------------
// The 32-frame kernel above issues 20 vector instructions per MFMA and re-reads the whole basis per 32 frames (1.5 GB per
// B = 256 launch, every k step a round trip to L2 for fragments that feed 3 MFMAs).  Here a wave owns ONE pair (cos / sin of 32
// bins) for two frame tiles: a k step is 4 fragment loads (register double-buffered, one running pointer) feeding 12 MFMAs, the
// basis is streamed once per 64 frames, and eight waves cover the 256 bins the mel bank reads (the Nyquist bin carries no mel
// weight when fmax = sr / 2; launch_fbank checks `mel_max_bin` and keeps the kernel above otherwise).  The power tile goes to
// LDS one 32-frame tile at a time (33 KB), so two workgroups share a CU.  |X|^2 is re^2 + im^2 here (the exact path keeps the
// reference's sqrt-then-square rounding; on a bf16 handle the features are rounded to 2^-8 right after).
constexpr int F2_FRAMES = 64;
constexpr int F2_THREADS = 512;
constexpr int F2_PT_STRIDE = 257;

// MODE 0: exact fp32 MFMA; 1: bf16x3 (two-way split, three products); 2: bf16x6 — the EXACT three-way split v = h + m + l of samples
// and basis and the six partial products above 2^-26 (h h, h m, m h, h l, l h, m m): fp32-grade spectra at 6 x 32 = 192 matrix
// cycles per 16 taps against 8 x 64 = 512 for the fp32 MFMA (F32X3 handles; see asnorm_fused.hip for the same arithmetic)
template <int MODE>
__global__ __launch_bounds__(F2_THREADS, 4) void fbank64_kernel(FbankTables tb, const float* __restrict__ wav, int L, int T,
                                                                float* __restrict__ mel) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ns = (F2_FRAMES - 1) * tb.hop + tb.win_length;
    const int ns_pad = ((ns + 15) & ~15) + 16;
    constexpr bool SPLIT = MODE == 1, X6 = MODE == 2;
    bf16_t* yh = reinterpret_cast<bf16_t*>(smem);                  // SPLIT: [ns_pad] hi | [ns_pad] lo; X6: h | m | l; else [ns_pad] fp32
    bf16_t* yl = yh + ns_pad;
    bf16_t* y3 = yl + ns_pad;
    float* ys = reinterpret_cast<float*>(smem);
    float* pt = X6 ? reinterpret_cast<float*>(y3 + ns_pad) : ys + ns_pad;      // [32][257]
    float* melw = pt + 32 * F2_PT_STRIDE;

    const int b = blockIdx.y;
    const int f0 = blockIdx.x * F2_FRAMES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* __restrict__ x = wav + (int64_t)b * L;
    const bool two = f0 + 32 < T;                                  // the last tile of an utterance may hold one frame tile only

    for (int i = tid; i < tb.n_melw; i += F2_THREADS) melw[i] = tb.mel_w[i];
    const int j0 = f0 * tb.hop + tb.lpad - tb.n_fft / 2;
    const float coef = tb.preemph;
    const bool interior = j0 >= 1 && j0 + ns_pad <= L;             // no reflection anywhere in the tile
    for (int i0 = tid; i0 < ns_pad; i0 += 4 * F2_THREADS) {
        float v[4], prev[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * F2_THREADS;
            int jj = j0 + i;
            if (!interior) {
                jj = jj < 0 ? -jj : jj;
                jj = jj >= L ? 2 * (L - 1) - jj : jj;
                jj = max(0, min(jj, L - 1));
            }
            const bool in = i < ns_pad;
            v[u] = in ? x[in ? jj : 0] : 0.0f;
            prev[u] = (in && coef >= 0.0f) ? x[jj == 0 ? 1 : jj - 1] : 0.0f;   // F.pad(reflect,(1,0)): x[-1] := x[1]
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * F2_THREADS;
            if (i < ns_pad) {
                float w = v[u];
                if (coef >= 0.0f) w = __fadd_rn(__fmul_rn(-coef, prev[u]), w);
                if (SPLIT || X6) {
                    const bf16_t hi = static_cast<bf16_t>(w);
                    const float r1 = w - static_cast<float>(hi);
                    const bf16_t mi = static_cast<bf16_t>(r1);
                    yh[i] = hi;
                    yl[i] = mi;
                    if (X6) y3[i] = static_cast<bf16_t>(r1 - static_cast<float>(mi));
                } else {
                    ys[i] = w;
                }
            }
        }
    }
    __syncthreads();

    const int r = lane & 31, h = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][c][e] = 0.0f;
    if (MODE == 0) {                                               // exact fp32 MFMA, same tap order as the 32-frame kernel
        const int qstride = tb.n_pairs * 2 * 64;
        const f32x4* __restrict__ bs = reinterpret_cast<const f32x4*>(tb.basis) + wave * 2 * 64 + lane;
        const float* a0 = ys + r * tb.hop + 4 * h;
        const float* a1 = a0 + 32 * tb.hop;
        f32x4 n0 = bs[0], n1 = bs[64];
        for (int q = 0; q < tb.n_q; ++q) {
            const f32x4 c0 = n0, c1 = n1;
            const int qn = min(q + 1, tb.n_q - 1) * qstride;
            n0 = bs[qn]; n1 = bs[qn + 64];
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(a0 + 8 * q);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0[j], c0[j], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0[j], c1[j], acc[0][1], 0, 0, 0);
            }
            if (two) {
                const f32x4 x1 = *reinterpret_cast<const f32x4*>(a1 + 8 * q);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1[j], c0[j], acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1[j], c1[j], acc[1][1], 0, 0, 0);
                }
            }
        }
    } else if (X6) {
        const int kstride = tb.n_pairs * 2 * 64;
        const bf16x8* __restrict__ bh = reinterpret_cast<const bf16x8*>(tb.basis_hi) + wave * 2 * 64 + lane;
        const bf16x8* __restrict__ bm = reinterpret_cast<const bf16x8*>(tb.basis_lo) + wave * 2 * 64 + lane;      // (the two-way "lo" IS m)
        const bf16x8* __restrict__ bl = reinterpret_cast<const bf16x8*>(tb.basis_l3) + wave * 2 * 64 + lane;
        const bf16_t* a0 = yh + r * tb.hop + 8 * h;
        for (int kk = 0; kk < tb.n_k16; ++kk) {
            const int kn = kk * kstride;
            const bf16x8 ch0 = bh[kn], ch1 = bh[kn + 64], cm0 = bm[kn], cm1 = bm[kn + 64], cl0 = bl[kn], cl1 = bl[kn + 64];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                if (mt == 1 && !two) break;
                const bf16_t* ap = a0 + mt * 32 * tb.hop + 16 * kk;
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(ap);
                const bf16x8 am = *reinterpret_cast<const bf16x8*>(ap + ns_pad);
                const bf16x8 al = *reinterpret_cast<const bf16x8*>(ap + 2 * ns_pad);
                acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, ch0, acc[mt][0], 0, 0, 0);      // small terms first
                acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, ch1, acc[mt][1], 0, 0, 0);
                acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, cl0, acc[mt][0], 0, 0, 0);
                acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, cl1, acc[mt][1], 0, 0, 0);
                acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, cm0, acc[mt][0], 0, 0, 0);
                acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, cm1, acc[mt][1], 0, 0, 0);
                acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, ch0, acc[mt][0], 0, 0, 0);
                acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, ch1, acc[mt][1], 0, 0, 0);
                acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, cm0, acc[mt][0], 0, 0, 0);
                acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, cm1, acc[mt][1], 0, 0, 0);
                acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, ch0, acc[mt][0], 0, 0, 0);
                acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, ch1, acc[mt][1], 0, 0, 0);
            }
        }
    } else {
        const int kstride = tb.n_pairs * 2 * 64;                                        // fragments per k step
        const bf16x8* __restrict__ bh = reinterpret_cast<const bf16x8*>(tb.basis_hi) + wave * 2 * 64 + lane;
        const bf16x8* __restrict__ bl = reinterpret_cast<const bf16x8*>(tb.basis_lo) + wave * 2 * 64 + lane;
        const bf16_t* a0 = yh + r * tb.hop + 8 * h;
        const bf16_t* a1 = a0 + 32 * tb.hop;
        bf16x8 nh0 = bh[0], nh1 = bh[64], nl0 = bl[0], nl1 = bl[64];
        for (int kk = 0; kk < tb.n_k16; ++kk) {
            const bf16x8 ch0 = nh0, ch1 = nh1, cl0 = nl0, cl1 = nl1;
            const int kn = min(kk + 1, tb.n_k16 - 1) * kstride;
            nh0 = bh[kn]; nh1 = bh[kn + 64]; nl0 = bl[kn]; nl1 = bl[kn + 64];
            const bf16x8 ah0 = *reinterpret_cast<const bf16x8*>(a0 + 16 * kk);
            const bf16x8 al0 = *reinterpret_cast<const bf16x8*>(a0 + ns_pad + 16 * kk);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al0, ch0, acc[0][0], 0, 0, 0);   // small terms first
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al0, ch1, acc[0][1], 0, 0, 0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, cl0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, cl1, acc[0][1], 0, 0, 0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, ch0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, ch1, acc[0][1], 0, 0, 0);
            if (two) {
                const bf16x8 ah1 = *reinterpret_cast<const bf16x8*>(a1 + 16 * kk);
                const bf16x8 al1 = *reinterpret_cast<const bf16x8*>(a1 + ns_pad + 16 * kk);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al1, ch0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al1, ch1, acc[1][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, cl0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, cl1, acc[1][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, ch0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, ch1, acc[1][1], 0, 0, 0);
            }
        }
    }

#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        if (mt == 1 && !two) break;
        if (mt) __syncthreads();                                   // the previous tile's mel sums have read pt
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float re = acc[mt][0][e], im = acc[mt][1][e];
            const int i = (e & 3) + 8 * (e >> 2) + 4 * h;
            if (SPLIT) {
                pt[i * F2_PT_STRIDE + wave * 32 + r] = fmaf(re, re, im * im);
            } else {
                const float mag = __fsqrt_rn(__fadd_rn(__fmul_rn(re, re), __fmul_rn(im, im)));   // 'Magnitude'
                pt[i * F2_PT_STRIDE + wave * 32 + r] = __fmul_rn(mag, mag);                      // ** 2.0
            }
        }
        __syncthreads();
        for (int idx = tid; idx < 32 * tb.n_mels; idx += F2_THREADS) {
            const int i = idx & 31, m = idx >> 5;
            const int f = f0 + 32 * mt + i;
            const int st = tb.mel_start[m], ln = tb.mel_len[m];
            const float* w = melw + tb.mel_off[m];
            const float* prow = pt + i * F2_PT_STRIDE + st;
            float sacc = 0.0f;
            for (int k = 0; k < ln; ++k) sacc = fmaf(w[k], prow[k], sacc);
            if (f < T) mel[((int64_t)b * tb.n_mels + m) * T + f] = sacc;
        }
    }
}

// ---- round 6: the FUSED front-end of bf16 handles --------------------------------------------------------------------------------
// waveform -> pre-emphasis -> windowed DFT -> power -> mel -> log(. + 1e-6) in ONE kernel that leaves frame-major log-mel rows and
// per-tile column sums; a second, streaming kernel subtracts the per-utterance time mean and writes the 16-bit operand of blocks.0.
// (Round 5: fbank64 106 us + prologue_stats 12 + prologue_apply 21 at B = 256, the fp32 mel (B, 80, T) written and read back twice.)
//
// The DFT uses the window's symmetry.  The periodic Hamming window of 200 taps sits at samples 156 .. 355 of the 512-sample frame,
// symmetric about sample 256 (tap 100), where cos(2 pi k s / 512) is even and sin is odd in m = s - 256 (up to the sign (-1)^k, which
// the power does not see).  So
//     Re X_k = sum_{m = 0..100} w_m cos(2 pi k m / 512) e_m,   e_0 = y_c, e_m = y_{c+m} + y_{c-m} (m < 100), e_100 = y_{c-100}
//     Im X_k = sum_{m = 1..100} w_m sin(2 pi k m / 512) o_m,   o_m = y_{c+m} - y_{c-m} (m < 100),            o_100 = -y_{c-100}
// (y = pre-emphasised samples, c = the frame's tap 100; tap 0 has no partner: slot 100): two products of K = 101 instead of one of
// K = 200 for both — HALF the matrix work and half the basis bytes streamed from L2 — at the price of an im2col-like operand:
// e and o per frame, split into bf16 hi | lo parts, 53 KB of LDS for 64 frames (rows of 104 values = 208 bytes: a row's 16-byte chunks
// land on 16 different bank groups for 16 consecutive rows; the seventh k step of 16 reads its upper half from the next row's first
// eight values, which meet zero basis rows).  Products as in fbank64<1>: hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_bf16.
constexpr int FF_FRAMES = 64;
constexpr int FF_THREADS = 512;
constexpr int FF_K = 104;                       // operand row: m = 0 .. 103 (101 used)
constexpr int FF_ROWB = FF_K * 2;               // 208 bytes
constexpr int FF_PLANE = FF_FRAMES * FF_ROWB;   // one of the four planes: e hi | e lo | o hi | o lo
constexpr int FF_ABYTES = 4 * FF_PLANE + 16;    // + 16 zero bytes behind the last row
constexpr int FF_KSTEPS = 7;
constexpr int FF_PT_STRIDE = 257;

__global__ __launch_bounds__(FF_THREADS, 4) void fbank_fused_kernel(FbankTables tb, const float* __restrict__ wav, int L, int T, int log_input,
                                                                    float* __restrict__ logmel, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ns = (FF_FRAMES - 1) * tb.hop + tb.win_length;
    const int ns_pad = ((ns + 15) & ~15) + 16;
    char* abuf = smem;                                             // [4][64][208 B]
    float* ys = reinterpret_cast<float*>(smem + FF_ABYTES);        // [ns_pad] pre-emphasised samples
    float* melw = ys + ns_pad;                                     // packed mel weights (+ 16 floats of slack: the unrolled sums read past a filter)
    int* mtab = reinterpret_cast<int*>(melw + tb.n_melw + 16);     // [3][n_mels]: first bin, length, weight offset of every filter
    float* pt = reinterpret_cast<float*>(smem);                    // [32][257] power tile: aliases the operand once the MFMAs are done
    float* lm = pt + 32 * FF_PT_STRIDE;                            // [32][n_mels + 1] log-mel tile
    const int lms = tb.n_mels + 1;

    const int b = blockIdx.y, tile = blockIdx.x;
    const int f0 = tile * FF_FRAMES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* __restrict__ x = wav + (int64_t)b * L;
    const bool two = f0 + 32 < T;

    for (int i = tid; i < tb.n_melw + 16; i += FF_THREADS) melw[i] = i < tb.n_melw ? tb.mel_w[i] : 0.0f;
    for (int i = tid; i < 3 * tb.n_mels; i += FF_THREADS) {
        const int k = i / tb.n_mels, m = i - k * tb.n_mels;
        mtab[i] = k == 0 ? tb.mel_start[m] : k == 1 ? tb.mel_len[m] : tb.mel_off[m];
    }
    if (tid < 4) reinterpret_cast<uint32_t*>(abuf + 4 * FF_PLANE)[tid] = 0u;
    // ---- 1. pre-emphasised, reflect-padded samples -> LDS (as fbank64) -----------------------------------------------------------
    const int j0 = f0 * tb.hop + tb.lpad - tb.n_fft / 2;
    const float coef = tb.preemph;
    const bool interior = j0 >= 1 && j0 + ns_pad <= L;
    if (interior && (((reinterpret_cast<uintptr_t>(x) >> 2) + (uintptr_t)j0) & 3) == 0 && coef >= 0.0f) {
        // no reflection anywhere in the tile and 16-byte aligned samples: every thread's loads (a float4 and the sample in front of it,
        // three times) go out in ONE batch
        const int nv = ns_pad >> 2;
        f32x4 v[3];
        float pv[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int q4 = tid + u * FF_THREADS;
            const bool in = q4 < nv && !(tb.ff_abl & 1);
            v[u] = in ? *reinterpret_cast<const f32x4*>(x + j0 + 4 * q4) : f32x4{0.f, 0.f, 0.f, 0.f};
            pv[u] = in ? x[j0 + 4 * q4 - 1] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int q4 = tid + u * FF_THREADS;
            if (q4 < nv) {
                f32x4 w;
                w[0] = __fadd_rn(__fmul_rn(-coef, pv[u]), v[u][0]);
                w[1] = __fadd_rn(__fmul_rn(-coef, v[u][0]), v[u][1]);
                w[2] = __fadd_rn(__fmul_rn(-coef, v[u][1]), v[u][2]);
                w[3] = __fadd_rn(__fmul_rn(-coef, v[u][2]), v[u][3]);
                *reinterpret_cast<f32x4*>(ys + 4 * q4) = w;
            }
        }
    } else
    for (int i0 = tid; i0 < ns_pad; i0 += 4 * FF_THREADS) {
        float v[4], prev[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * FF_THREADS;
            int jj = j0 + i;
            if (!interior) {
                jj = jj < 0 ? -jj : jj;
                jj = jj >= L ? 2 * (L - 1) - jj : jj;
                jj = max(0, min(jj, L - 1));
            }
            const bool in = i < ns_pad && !(tb.ff_abl & 1);
            v[u] = in ? x[in ? jj : 0] : 0.0f;
            prev[u] = (in && coef >= 0.0f) ? x[jj == 0 ? 1 : jj - 1] : 0.0f;   // F.pad(reflect,(1,0)): x[-1] := x[1]
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * FF_THREADS;
            if (i < ns_pad) {
                float w = v[u];
                if (coef >= 0.0f) w = __fadd_rn(__fmul_rn(-coef, prev[u]), w);
                ys[i] = w;
            }
        }
    }
    __syncthreads();

    // ---- 2. the symmetric operand: e | o of every frame, bf16 hi | lo ---------------------------------------------------------------
    const int half = tb.win_length >> 1;                           // 100: the tap the window is symmetric about
    for (int wk = tid; wk < ((tb.ff_abl & 2) ? 0 : FF_FRAMES * (FF_K / 8)); wk += FF_THREADS) {
        const int i = wk / (FF_K / 8), j = wk - i * (FF_K / 8);
        const int c = i * tb.hop + half;
        const f32x4 fa = *reinterpret_cast<const f32x4*>(ys + c + 8 * j), fb = *reinterpret_cast<const f32x4*>(ys + c + 8 * j + 4);
        // y[c - 8 j - u], u = 1 .. 8 from two aligned reads (the chunk m = 96 .. 103 of frame 0 reads four floats below ys: masked)
        const f32x4 ra = *reinterpret_cast<const f32x4*>(ys + c - 8 * j - 8), rb = *reinterpret_cast<const f32x4*>(ys + c - 8 * j - 4);
        const float r0 = ys[c - 8 * j];
        bf16x8 eh, el, oh, ol;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int m = 8 * j + u;
            const float f = u < 4 ? fa[u] : fb[u - 4];
            const float r = u == 0 ? r0 : (u <= 4 ? rb[4 - u] : ra[8 - u]);
            float e = f + r, o = f - r;
            if (m == 0) { e = f; o = 0.0f; }
            if (m == half) { e = r; o = -r; }
            if (m > half) { e = 0.0f; o = 0.0f; }
            const bf16_t e1 = static_cast<bf16_t>(e), o1 = static_cast<bf16_t>(o);
            eh[u] = e1; el[u] = static_cast<bf16_t>(e - static_cast<float>(e1));
            oh[u] = o1; ol[u] = static_cast<bf16_t>(o - static_cast<float>(o1));
        }
        char* row = abuf + i * FF_ROWB + j * 16;
        *reinterpret_cast<bf16x8*>(row) = eh;
        *reinterpret_cast<bf16x8*>(row + FF_PLANE) = el;
        *reinterpret_cast<bf16x8*>(row + 2 * FF_PLANE) = oh;
        *reinterpret_cast<bf16x8*>(row + 3 * FF_PLANE) = ol;
    }
    __syncthreads();

    // ---- 3. the two products: wave = one group of 32 bins (cos and sin), both frame tiles -------------------------------------------
    const int r = lane & 31, h = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][c][e] = 0.0f;
    {
        const int kstride = 8 * 2 * 64;                            // fragments per k step: [pair][part][lane]
        const bf16x8* __restrict__ bh = reinterpret_cast<const bf16x8*>(tb.sym_hi) + wave * 2 * 64 + lane;
        const bf16x8* __restrict__ bl = reinterpret_cast<const bf16x8*>(tb.sym_lo) + wave * 2 * 64 + lane;
        const char* a0 = abuf + r * FF_ROWB + 16 * h;
        bf16x8 nh0 = bh[0], nh1 = bh[64], nl0 = bl[0], nl1 = bl[64];
#pragma unroll
        for (int kk = 0; kk < FF_KSTEPS; ++kk) {
            if (tb.ff_abl & 4) break;
            const bf16x8 ch = nh0, sh = nh1, cl = nl0, sl = nl1;
            const int kn = min(kk + 1, FF_KSTEPS - 1) * kstride;
            nh0 = bh[kn]; nh1 = bh[kn + 64]; nl0 = bl[kn]; nl1 = bl[kn + 64];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                if (mt == 1 && !two) break;
                const char* ap = a0 + mt * 32 * FF_ROWB + kk * 32;
                const bf16x8 eh = *reinterpret_cast<const bf16x8*>(ap);
                const bf16x8 el = *reinterpret_cast<const bf16x8*>(ap + FF_PLANE);
                const bf16x8 oh = *reinterpret_cast<const bf16x8*>(ap + 2 * FF_PLANE);
                const bf16x8 ol = *reinterpret_cast<const bf16x8*>(ap + 3 * FF_PLANE);
                acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(el, ch, acc[mt][0], 0, 0, 0);   // small terms first
                acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ol, sh, acc[mt][1], 0, 0, 0);
                acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(eh, cl, acc[mt][0], 0, 0, 0);
                acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(oh, sl, acc[mt][1], 0, 0, 0);
                acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(eh, ch, acc[mt][0], 0, 0, 0);
                acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(oh, sh, acc[mt][1], 0, 0, 0);
            }
        }
    }
    __syncthreads();                                               // the operand is dead: its space becomes the power / log-mel tiles

    // ---- 4. power -> mel -> log -> frame-major rows + column sums ------------------------------------------------------------------
    // (outputs idx = tid + 512 j: frame idx & 31, filter idx >> 5 — the same filters in both passes, so a thread keeps its column sums)
    constexpr int FF_MAXOUT = 5;                                   // ceil(32 * 80 / 512); larger banks loop
    float csum[FF_MAXOUT];
#pragma unroll
    for (int j = 0; j < FF_MAXOUT; ++j) csum[j] = 0.0f;
    const int* mst = mtab;
    const int* mln = mtab + tb.n_mels;
    const int* mof = mtab + 2 * tb.n_mels;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        if (mt == 1 && !two) break;
        if (mt) __syncthreads();                                   // the previous pass has read pt and lm
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float re = acc[mt][0][e], im = acc[mt][1][e];
            const int i = (e & 3) + 8 * (e >> 2) + 4 * h;
            pt[i * FF_PT_STRIDE + wave * 32 + r] = fmaf(re, re, im * im);
        }
        __syncthreads();
        const int fbase = f0 + 32 * mt;
#pragma unroll
        for (int j = 0; j < FF_MAXOUT; ++j) {
            const int idx = tid + j * FF_THREADS;
            if (idx >= 32 * tb.n_mels || (tb.ff_abl & 8)) break;
            const int i = idx & 31, m = idx >> 5;
            const int st = mst[m], ln = mln[m];
            const float* w = melw + mof[m];
            const float* prow = pt + i * FF_PT_STRIDE + st;
            float s0 = 0.0f, s1 = 0.0f;
            int k = 0;
            for (; k + 1 < ln; k += 2) { s0 = fmaf(w[k], prow[k], s0); s1 = fmaf(w[k + 1], prow[k + 1], s1); }
            if (k < ln) s0 = fmaf(w[k], prow[k], s0);
            const float sacc = s0 + s1;
            const float v = log_input ? __logf(sacc + 1e-6f) : sacc;
            lm[i * lms + m] = v;
            // this tile's column sum over its valid frames: a fixed tree over the 32 lanes that share the filter (the same bits every run)
            float cv = fbase + i < T ? v : 0.0f;
#pragma unroll
            for (int o = 16; o >= 1; o >>= 1) cv += __shfl_xor(cv, o, 64);
            csum[j] += cv;
        }
        __syncthreads();
        for (int idx = tid; idx < 32 * tb.n_mels; idx += FF_THREADS) {
            const int f = idx / tb.n_mels, m = idx - f * tb.n_mels;
            if (fbase + f < T && !(tb.ff_abl & 16)) logmel[((int64_t)b * T + fbase + f) * tb.n_mels + m] = lm[f * lms + m];
        }
    }
    if ((tid & 31) == 0) {
#pragma unroll
        for (int j = 0; j < FF_MAXOUT; ++j) {
            const int m = (tid + j * FF_THREADS) >> 5;
            if (m < tb.n_mels) partial[((int64_t)b * gridDim.x + tile) * tb.n_mels + m] = csum[j];
        }
    }
}

// log-mel rows (B, T, n_mels) fp32 - per-utterance time mean (from the tiles' column sums, added in tile order) -> 16-bit operand rows
template <typename T_>
__global__ __launch_bounds__(256) void fbank_norm_kernel(const float* __restrict__ logmel, const float* __restrict__ partial, T_* __restrict__ out,
                                                         int T, int n_mels, int ntiles, int sub_mean) {
    __shared__ float mean[256];
    const int b = blockIdx.y, t0 = blockIdx.x * 64;
    if ((int)threadIdx.x < n_mels) {
        float s = 0.0f;
        for (int k = 0; k < ntiles; ++k) s += partial[((int64_t)b * ntiles + k) * n_mels + threadIdx.x];
        mean[threadIdx.x] = sub_mean ? s / (float)T : 0.0f;
    }
    __syncthreads();
    const int nq = n_mels >> 2;
    for (int q = threadIdx.x; q < 64 * nq; q += 256) {
        const int f = q / nq, mq = q - f * nq;
        const int t = t0 + f;
        if (t >= T) break;
        const int64_t o = ((int64_t)b * T + t) * n_mels + 4 * mq;
        const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(logmel + o));
        T_ y[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) y[u] = from_f32<T_>(v[u] - mean[4 * mq + u]);
        if (sizeof(T_) == 2) {
            uint2 pk;
            pk.x = (uint32_t)__builtin_bit_cast(uint16_t, y[0]) | ((uint32_t)__builtin_bit_cast(uint16_t, y[1]) << 16);
            pk.y = (uint32_t)__builtin_bit_cast(uint16_t, y[2]) | ((uint32_t)__builtin_bit_cast(uint16_t, y[3]) << 16);
            *reinterpret_cast<uint2*>(out + o) = pk;
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) out[o + u] = y[u];
        }
    }
}

// per (b, mel): mean of log(x + 1e-6) over T (and biased variance of the normalised signal when
// instance norm is on).  stats[(b*n_mels + m)*2 + {0,1}] = {shift, scale}: y = (v - shift) * scale.
__global__ __launch_bounds__(256) void prologue_stats_kernel(const float* __restrict__ feat, float* __restrict__ stats,
                                                             int rows, int T, int log_input, int inorm) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const float* __restrict__ x = feat + (int64_t)row * T;
    float s = 0.0f;
    for (int t = lane; t < T; t += 64) {
        float v = x[t];
        if (log_input) v = logf(v + 1e-6f);
        s += v;
    }
    const float mean = wave_sum(s) / (float)T;
    float shift = log_input ? mean : 0.0f, scale = 1.0f;
    if (inorm) {
        // InstanceNorm1d over t of u = (v - shift): mean_u = mean - shift, var biased
        const float mu = mean - shift;
        float q = 0.0f;
        for (int t = lane; t < T; t += 64) {
            float v = x[t];
            if (log_input) v = logf(v + 1e-6f);
            const float d = (v - shift) - mu;
            q += d * d;
        }
        const float var = wave_sum(q) / (float)T;
        shift = shift + mu;
        scale = 1.0f / sqrtf(var + 1e-5f);
    }
    if (lane == 0) { stats[2 * row] = shift; stats[2 * row + 1] = scale; }
}

// (B, n_mels, T) -> (B, T, n_mels) with log / shift / scale / affine applied; 32-frame LDS transpose tile.
template <typename T_>
__global__ __launch_bounds__(256) void prologue_apply_kernel(const float* __restrict__ feat, const float* __restrict__ stats,
                                                             T_* __restrict__ out, int n_mels, int T, int log_input,
                                                             const float* __restrict__ in_w, const float* __restrict__ in_b,
                                                             uint32_t* __restrict__ status, volatile uint32_t* __restrict__ host_flag, float limit) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* tile = reinterpret_cast<float*>(smem);      // [n_mels][33]
    const int b = blockIdx.y, t0 = blockIdx.x * 32;
    bool bad = false;
    for (int idx = threadIdx.x; idx < n_mels * 32; idx += 256) {
        const int i = idx & 31, m = idx >> 5;
        const int t = t0 + i;
        float v = 0.0f;
        if (t < T) {
            v = feat[((int64_t)b * n_mels + m) * T + t];
            if (log_input) v = logf(v + 1e-6f);
            const float sh = stats[2 * (b * n_mels + m)], sc = stats[2 * (b * n_mels + m) + 1];
            v = (v - sh) * sc;
            if (in_w) v = v * in_w[m] + in_b[m];
        }
        tile[m * 33 + i] = v;
        bad |= !(fabsf(v) <= limit);
    }
    // range guard of the split (F32X3) handles: the network's input leaves for half-precision hi | lo planes, whose hi part saturates at
    // 65504 — a value beyond `limit` (or a NaN) raises bit 1 of status[0], is counted in status[2] and sets the host-visible flag
    if (status) {
        const unsigned long long mb = __ballot(bad);
        if (mb && (threadIdx.x & 63) == 0) {
            atomicOr(status, 2u);
            atomicAdd(status + 2, (uint32_t)__popcll(mb));
            __threadfence_system();
            *host_flag = 1u;
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < n_mels * 32; idx += 256) {
        const int m = idx % n_mels, i = idx / n_mels;
        const int t = t0 + i;
        if (t < T) out[((int64_t)b * T + t) * n_mels + m] = from_f32<T_>(tile[m * 33 + i]);
    }
}

}  // namespace

hipError_t launch_fbank(const FbankTables& tb, const float* wav, int B, int L, int T, float* mel, hipStream_t stream) {
    if (tb.win_length % 8 != 0 || tb.hop % 4 != 0 || tb.n_pairs > 3 * FB_PAIRS_PER_WAVE ||
        tb.n_bins > 32 * tb.n_pairs || tb.n_q * 8 != tb.win_length || L < tb.n_fft || B <= 0)
        return hipErrorInvalidValue;
    const int ns = (FB_FRAMES - 1) * tb.hop + tb.win_length;
    const int ns_pad = ((ns + 15) & ~15) + 16;
    const size_t lds = (size_t)(ns_pad + FB_FRAMES * FB_PT_STRIDE + tb.n_melw) * sizeof(float);
    dim3 grid((T + FB_FRAMES - 1) / FB_FRAMES, B), block(FB_THREADS);
    if (tb.split_bf16 && (!tb.basis_hi || !tb.basis_lo || tb.hop % 8 != 0)) return hipErrorInvalidValue;
    // 64-frame kernel: needs the mel bank to stop below the Nyquist bin (eight pairs = 256 bins) and its tile in 64 KiB of LDS
    const int ns2 = (F2_FRAMES - 1) * tb.hop + tb.win_length;
    const int ns2_pad = ((ns2 + 15) & ~15) + 16;
    const bool x6 = tb.split6 && tb.basis_hi && tb.basis_lo && tb.basis_l3 && tb.hop % 8 == 0;
    const size_t lds2 = (size_t)(32 * F2_PT_STRIDE + tb.n_melw) * sizeof(float) + (size_t)ns2_pad * (x6 ? 6 : 4);
    if (tb.mel_max_bin < 256 && tb.n_pairs >= 8 && lds2 <= 80 * 1024 && !tb.force32) {
        dim3 grid2((T + F2_FRAMES - 1) / F2_FRAMES, B);
        if (x6) {
            static DeviceOnce attr6;
            if (hipError_t e = set_max_dynamic_lds(attr6, reinterpret_cast<const void*>(fbank64_kernel<2>), 80 * 1024)) return e;
            hipLaunchKernelGGL(fbank64_kernel<2>, grid2, dim3(F2_THREADS), lds2, stream, tb, wav, L, T, mel);
        } else if (tb.split_bf16)
            hipLaunchKernelGGL(fbank64_kernel<1>, grid2, dim3(F2_THREADS), lds2, stream, tb, wav, L, T, mel);
        else
            hipLaunchKernelGGL(fbank64_kernel<0>, grid2, dim3(F2_THREADS), lds2, stream, tb, wav, L, T, mel);
        return hipGetLastError();
    }
    if (tb.split_bf16)
        hipLaunchKernelGGL(fbank_kernel<true>, grid, block, lds, stream, tb, wav, L, T, mel);
    else
        hipLaunchKernelGGL(fbank_kernel<false>, grid, block, lds, stream, tb, wav, L, T, mel);
    return hipGetLastError();
}

bool fbank_fused_supported(const FbankTables& tb, int L) {
    return tb.sym_hi && tb.sym_lo && tb.n_fft == 512 && tb.win_length == 200 && tb.hop == 80 && tb.lpad == 156 && tb.mel_max_bin < 256 &&
           tb.n_mels % 4 == 0 && tb.n_mels <= 128 && tb.n_pairs >= 8 && L >= tb.n_fft;
}

// wav (B, L) -> out (B * T, n_mels) bf16 rows = log(mel + 1e-6) - time mean (log_input) or the mel power itself; `logmel` (B, T, n_mels)
// fp32 and `partial` (B, ceil(T / 64), n_mels) fp32 are scratch
hipError_t launch_fbank_fused(const FbankTables& tb, const float* wav, int B, int L, int T, int log_input, float* logmel, float* partial,
                              void* out, hipStream_t stream) {
    if (!fbank_fused_supported(tb, L) || B <= 0) return hipErrorInvalidValue;
    const int ns = (FF_FRAMES - 1) * tb.hop + tb.win_length;
    const int ns_pad = ((ns + 15) & ~15) + 16;
    const size_t lds = (size_t)FF_ABYTES + (size_t)(ns_pad + tb.n_melw + 16 + 3 * tb.n_mels) * sizeof(float);
    if (32 * tb.n_mels > 5 * FF_THREADS) return hipErrorInvalidValue;                 // (FF_MAXOUT outputs per thread and pass)
    if (lds > 80 * 1024 || (size_t)(32 * FF_PT_STRIDE + 32 * (tb.n_mels + 1)) * 4 > (size_t)4 * FF_PLANE) return hipErrorInvalidValue;
    static DeviceOnce attr;
    if (hipError_t e = set_max_dynamic_lds(attr, reinterpret_cast<const void*>(fbank_fused_kernel), 80 * 1024)) return e;
    const int ntiles = (T + FF_FRAMES - 1) / FF_FRAMES;
    hipLaunchKernelGGL(fbank_fused_kernel, dim3(ntiles, B), dim3(FF_THREADS), lds, stream, tb, wav, L, T, log_input, logmel, partial);
    if (!(tb.ff_abl & 32))
        hipLaunchKernelGGL(fbank_norm_kernel<bf16_t>, dim3((T + 63) / 64, B), dim3(256), 0, stream, logmel, partial, reinterpret_cast<bf16_t*>(out), T,
                           tb.n_mels, ntiles, log_input);
    return hipGetLastError();
}

hipError_t launch_prologue(const float* feat, void* out, bool out_bf16, int B, int n_mels, int T, int log_input,
                           const float* in_w, const float* in_b, float* stats, hipStream_t stream, uint32_t* status, uint32_t* host_flag, float limit) {
    const int rows = B * n_mels;
    hipLaunchKernelGGL(prologue_stats_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, feat, stats, rows, T,
                       log_input, in_w != nullptr ? 1 : 0);
    dim3 grid((T + 31) / 32, B);
    const size_t lds = (size_t)n_mels * 33 * sizeof(float);
    if (out_bf16)
        hipLaunchKernelGGL(prologue_apply_kernel<bf16_t>, grid, dim3(256), lds, stream, feat, stats,
                           reinterpret_cast<bf16_t*>(out), n_mels, T, log_input, in_w, in_b, status, host_flag, limit);
    else
        hipLaunchKernelGGL(prologue_apply_kernel<float>, grid, dim3(256), lds, stream, feat, stats,
                           reinterpret_cast<float*>(out), n_mels, T, log_input, in_w, in_b, status, host_flag, limit);
    return hipGetLastError();
}

}  // namespace svhip
