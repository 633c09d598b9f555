// score.hip — trial scoring kernels (HBM-bound gathers + per-row top-k statistics).
//   l2norm        : F.normalize(p=2, dim=1)                              (reference src/model.py:421-423)
//   score_pairs   : | cos(E[a], E[b]) | with per-norm clamp 1e-5         (src/utils.py:163-164)
//   asnorm_pairs  : 0.5*((s-mu_a)/sd_a + (s-mu_b)/sd_b), s = E[a].E[b]  (src/utils.py:148-160)
//   topk_stats    : per row of a (rows x K) cohort-score slab: mean / population std of the `top`
//                   largest values                                       (src/utils.py:142-146)
// Pair kernels use 16 lanes per pair (4 pairs per wavefront): each lane loads float4s of both rows
// (whole-row gathers), partial dot products are reduced with 4 xor-shuffles.
#include "common.h"
#include "kernels.h"
#include "score_select.h"

namespace svhip {

namespace {

__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__global__ __launch_bounds__(256) void l2norm_kernel(float* __restrict__ E, int64_t N, int D) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    float* e = E + row * D;
    float s = 0.0f;
    for (int k = lane; k < D; k += 64) s = fmaf(e[k], e[k], s);
    s = wave_sum(s);
    const float inv = 1.0f / fmaxf(sqrtf(s), 1e-12f);
    for (int k = lane; k < D; k += 64) e[k] *= inv;
}

// MODE 0: |cos| with clamp; MODE 1: AS-norm of the raw dot product
template <int MODE>
__global__ __launch_bounds__(256) void pair_kernel(const float* __restrict__ E, int D, const int32_t* __restrict__ ia,
                                                   const int32_t* __restrict__ ib, int64_t P, const float* __restrict__ mu,
                                                   const float* __restrict__ sigma, float* __restrict__ out) {
    const int sub = threadIdx.x & 15;
    const int64_t p = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
    const bool ok = p < P;
    const int64_t a = ok ? ia[p] : 0, b = ok ? ib[p] : 0;
    const float* __restrict__ ea = E + a * D;
    const float* __restrict__ eb = E + b * D;
    float dab = 0.f, daa = 0.f, dbb = 0.f;
    if ((D & 3) == 0) {
        for (int k = sub * 4; k < D; k += 64) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(ea + k);
            const f32x4 y = *reinterpret_cast<const f32x4*>(eb + k);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dab = fmaf(x[j], y[j], dab);
                if (MODE == 0) { daa = fmaf(x[j], x[j], daa); dbb = fmaf(y[j], y[j], dbb); }
            }
        }
    } else {
        for (int k = sub; k < D; k += 16) {
            const float x = ea[k], y = eb[k];
            dab = fmaf(x, y, dab);
            if (MODE == 0) { daa = fmaf(x, x, daa); dbb = fmaf(y, y, dbb); }
        }
    }
    dab = group16_sum(dab);
    if (MODE == 0) { daa = group16_sum(daa); dbb = group16_sum(dbb); }
    if (ok && sub == 0) {
        if (MODE == 0) {
            out[p] = fabsf(dab / (fmaxf(sqrtf(daa), 1e-5f) * fmaxf(sqrtf(dbb), 1e-5f)));
        } else {
            out[p] = 0.5f * ((dab - mu[a]) / sigma[a] + (dab - mu[b]) / sigma[b]);
        }
    }
}

// ---- whole-trial forms over the crops of two files: F (n_files, n_crops, D), 16 lanes per trial ---------------------------
//   MODE 0: mean_i | cos(R_i, C_i) |  (crop-aligned, per-norm clamp 1e-5)                        src/utils.py:163-164
//   MODE 1: mean_i || R_i - C_i + 1e-6 ||_2                                                      src/utils.py:167-169
//   MODE 2: - mean_{i,d} sqrt( sum_j (R[i,d] - C[j,d] + 1e-6)^2 )  — F.pairwise_distance of ref (n, D, 1) against com (1, D, n)
//           takes the 2-norm over the LAST axis of the broadcast (n, D, n) difference (src/model.py:425-431)
//   MODE 3: mean_i || R_i - C_i + 1e-6 ||_p for any p (pnorm_similarity's `p` argument, src/utils.py:167-169 -> F.pairwise_distance ->
//           torch.linalg.vector_norm): p = inf max |d|, p = -inf min |d|, p = 0 the count of non-zero d, p = 1 the sum of |d|,
//           otherwise (sum |d|^p)^(1/p) — negative p included, as torch evaluates it
template <int MODE>
__global__ __launch_bounds__(256) void trial_crops_kernel(const float* __restrict__ F, int n_crops, int D, const int32_t* __restrict__ ia,
                                                          const int32_t* __restrict__ ib, int64_t P, float* __restrict__ out, float pexp) {
    const int sub = threadIdx.x & 15;
    const int64_t p = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
    const bool ok = p < P;
    const int64_t a = ok ? ia[p] : 0, b = ok ? ib[p] : 0;
    const float* __restrict__ fa = F + a * n_crops * D;
    const float* __restrict__ fb = F + b * n_crops * D;
    float acc = 0.0f;
    if (MODE == 2) {
        for (int d = sub; d < D; d += 16)
            for (int i = 0; i < n_crops; ++i) {
                const float r = fa[(int64_t)i * D + d];
                float q = 0.0f;
                for (int j = 0; j < n_crops; ++j) { const float t = (r - fb[(int64_t)j * D + d]) + 1e-6f; q = fmaf(t, t, q); }
                acc += sqrtf(q);
            }
        acc = group16_sum(acc);
        if (ok && sub == 0) out[p] = -acc / ((float)n_crops * (float)D);
        return;
    }
    if (MODE == 3) {
        const bool pinf = pexp > 3.0e38f, ninf = pexp < -3.0e38f;
        for (int i = 0; i < n_crops; ++i) {
            const float* __restrict__ ea = fa + (int64_t)i * D;
            const float* __restrict__ eb = fb + (int64_t)i * D;
            float r = ninf ? __builtin_inff() : 0.0f;
            for (int k = sub; k < D; k += 16) {
                const float t = fabsf((ea[k] - eb[k]) + 1e-6f);
                if (pinf) r = fmaxf(r, t);
                else if (ninf) r = fminf(r, t);
                else if (pexp == 0.0f) r += t != 0.0f ? 1.0f : 0.0f;
                else if (pexp == 1.0f) r += t;
                else r += powf(t, pexp);
            }
            // the 16 lanes of a trial: xor shuffles inside the group
#pragma unroll
            for (int o = 8; o >= 1; o >>= 1) {
                const float u = __shfl_xor(r, o, 64);
                r = pinf ? fmaxf(r, u) : ninf ? fminf(r, u) : r + u;
            }
            acc += (pinf || ninf || pexp == 0.0f || pexp == 1.0f) ? r : powf(r, 1.0f / pexp);
        }
        if (ok && sub == 0) out[p] = acc / (float)n_crops;
        return;
    }
    for (int i = 0; i < n_crops; ++i) {
        const float* __restrict__ ea = fa + (int64_t)i * D;
        const float* __restrict__ eb = fb + (int64_t)i * D;
        float dab = 0.f, daa = 0.f, dbb = 0.f;
        for (int k = sub; k < D; k += 16) {
            const float x = ea[k], y = eb[k];
            if (MODE == 0) { dab = fmaf(x, y, dab); daa = fmaf(x, x, daa); dbb = fmaf(y, y, dbb); }
            else { const float t = (x - y) + 1e-6f; dab = fmaf(t, t, dab); }
        }
        dab = group16_sum(dab);
        if (MODE == 0) {
            daa = group16_sum(daa); dbb = group16_sum(dbb);
            acc += fabsf(dab / (fmaxf(sqrtf(daa), 1e-5f) * fmaxf(sqrtf(dbb), 1e-5f)));
        } else {
            acc += sqrtf(dab);
        }
    }
    if (ok && sub == 0) out[p] = acc / (float)n_crops;
}

// crop means: F (n_files, n_crops, D) -> out (n_files, D)   (np.mean over the crops: the AS-norm statement on crop means)
__global__ __launch_bounds__(256) void mean_crops_kernel(const float* __restrict__ F, int64_t n_files, int n_crops, int D, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_files * D) return;
    const int64_t f = i / D;
    const int d = (int)(i - f * D);
    float s = 0.0f;
    for (int c = 0; c < n_crops; ++c) s += F[(f * n_crops + c) * D + d];
    out[i] = s / (float)n_crops;
}

constexpr int TOPK_CAP = 512;            // candidate slots per row (8 per lane)

// Register-resident variant: one wavefront per row, the row lives in NV float4 registers per lane (K <= 256*NV),
// only the <= 512 compacted candidates touch LDS (2 KiB per wave), so many waves per CU hide the HBM latency of
// the slab read.  Same selection logic as topk_stats_kernel; the rare fallback is the exact 32-pass search
// over the registers.
template <int NV>
__global__ __launch_bounds__(256) void topk_stats_reg_kernel(const float* __restrict__ S, int64_t rows, int K, int ld, int top,
                                                             float* __restrict__ mu, float* __restrict__ sigma) {
    __shared__ uint32_t cand_all[4][TOPK_CAP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= rows) return;
    uint32_t* cand = cand_all[wave];
    const float* __restrict__ s = S + row * ld;
    u32x4 kv[NV];
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int k0 = (j * 64 + lane) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k0 < K) v = *reinterpret_cast<const f32x4*>(s + k0);            // ld % 4 == 0: a ragged tail reads row padding
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool in = k0 + e < K;
            kv[j][e] = in ? fkey(v[e]) : 0u;
            if (in) { s1 += v[e]; s2 = fmaf(v[e], v[e], s2); }
        }
    }
    float mean_out = 0.f, sd_out = 0.f;
    bool done = false;
    if (top <= 256 && K >= 4 * top) {
        s1 = wave_sum(s1); s2 = wave_sum(s2);
        const float m = s1 / (float)K;
        const float sd = sqrtf(fmaxf(s2 / (float)K - m * m, 0.0f));
        const float frac = 2.0f * (float)top / (float)K;
        const float tq = sqrtf(-2.0f * logf(fmaxf(frac, 1e-6f)));
        float z = tq - (2.30753f + 0.27061f * tq) / (1.0f + 0.99229f * tq + 0.04481f * tq * tq);
        for (int attempt = 0; attempt < 4 && !done; ++attempt) {
            const uint32_t tkey = fkey(m + z * sd);
            int base = 0;
#pragma unroll
            for (int j = 0; j < NV; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool pred = kv[j][e] > tkey;
                    const unsigned long long mask = __ballot(pred);
                    const int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
                    if (pred && pos < TOPK_CAP) cand[pos] = kv[j][e];
                    base += __popcll(mask);
                }
            if (base >= top && base <= TOPK_CAP) {
                uint32_t ck[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { const int idx = lane + 64 * j; ck[j] = idx < base ? cand[idx] : 0u; }
                select_stats<8>(ck, top, mean_out, sd_out);
                done = true;
            } else {
                z += (base < top) ? -0.6f : 0.5f;
            }
        }
    }
    if (!done) {
        uint32_t flat[NV * 4];
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) flat[j * 4 + e] = kv[j][e];
        select_stats<NV * 4>(flat, top, mean_out, sd_out);
    }
    if (lane == 0) { mu[row] = mean_out; sigma[row] = sd_out; }
}

// One wavefront per row.  The row (K fp32 scores, row stride ld) is staged in LDS as sortable keys.
// Fast path (top <= 256, K >= 4*top): a threshold tau = mean + z*std picked from the row's own first
// two moments keeps <= 512 candidates (expected ~2*top), which are compacted with wave ballots and
// selected exactly in registers; tau is re-aimed at most 3 times.  Anything else (small K, unlucky
// distributions) takes the exact 32-pass search over the whole row.  Both paths give the same answer.
__global__ __launch_bounds__(256) void topk_stats_kernel(const float* __restrict__ S, int64_t rows, int K, int ld, int Kp, int top,
                                                         int rows_per_wg, float* __restrict__ mu, float* __restrict__ sigma) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t* keys = reinterpret_cast<uint32_t*>(smem) + (size_t)wave * (Kp + TOPK_CAP);
    uint32_t* cand = keys + Kp;
    const int64_t row = (int64_t)blockIdx.x * rows_per_wg + wave;
    const bool active = (wave < rows_per_wg) && (row < rows);
    float s1 = 0.0f, s2 = 0.0f;
    if (active) {
        const float* __restrict__ s = S + row * ld;
        for (int k = lane; k < Kp; k += 64) {
            const float v = (k < K) ? s[k] : 0.0f;
            keys[k] = (k < K) ? fkey(v) : 0u;                                       // pad = smallest key
            if (k < K) { s1 += v; s2 = fmaf(v, v, s2); }
        }
    }
    __syncthreads();
    if (!active) return;
    const u32x4* k4 = reinterpret_cast<const u32x4*>(keys);
    const int n4 = Kp >> 2;
    float mean_out, sd_out;
    bool done = false;
    if (top <= 256 && K >= 4 * top) {
        s1 = wave_sum(s1); s2 = wave_sum(s2);
        const float m = s1 / (float)K;
        const float sd = sqrtf(fmaxf(s2 / (float)K - m * m, 0.0f));
        // aim for ~2*top candidates: upper-tail quantile of a normal with the row's moments
        const float frac = 2.0f * (float)top / (float)K;                         // <= 0.5
        const float tq = sqrtf(-2.0f * logf(fmaxf(frac, 1e-6f)));               // Abramowitz-Stegun 26.2.22 inverse normal tail
        float z = tq - (2.30753f + 0.27061f * tq) / (1.0f + 0.99229f * tq + 0.04481f * tq * tq);
        for (int attempt = 0; attempt < 4 && !done; ++attempt) {
            const uint32_t tkey = fkey(m + z * sd);
            int base = 0;
            for (int i = lane; i < n4 + 63 - ((n4 + 63) % 64) ; i += 64) {       // every lane runs the same trip count (ballots)
                const bool in = i < n4;
                const u32x4 v = in ? k4[i] : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool pred = v[j] > tkey;
                    const unsigned long long mask = __ballot(pred);
                    const int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
                    if (pred && pos < TOPK_CAP) cand[pos] = v[j];
                    base += __popcll(mask);
                }
            }
            if (base >= top && base <= TOPK_CAP) {
                uint32_t ck[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { const int idx = lane + 64 * j; ck[j] = idx < base ? cand[idx] : 0u; }
                select_stats<8>(ck, top, mean_out, sd_out);
                done = true;
            } else {
                z += (base < top) ? -0.6f : 0.5f;
            }
        }
    }
    if (!done) {
        uint32_t prefix = 0;
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t c = prefix | (1u << bit);
            int cnt = 0;
            for (int i = lane; i < n4; i += 64) {
                const u32x4 v = k4[i];
                cnt += (v[0] >= c) + (v[1] >= c) + (v[2] >= c) + (v[3] >= c);
            }
            if (wave_isum(cnt) >= top) prefix = c;
        }
        const float vth = fkey_inv(prefix);
        float sum = 0.0f;
        int cgt = 0;
        for (int i = lane; i < n4; i += 64) {
            const u32x4 v = k4[i];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (v[j] > prefix) { sum += fkey_inv(v[j]); ++cgt; }
        }
        sum = wave_sum(sum);
        cgt = wave_isum(cgt);
        const float nt = (float)(top - cgt);
        const float mean = (sum + nt * vth) / (float)top;
        float sq = 0.0f;
        for (int i = lane; i < n4; i += 64) {
            const u32x4 v = k4[i];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (v[j] > prefix) { const float d = fkey_inv(v[j]) - mean; sq = fmaf(d, d, sq); }
        }
        sq = wave_sum(sq);
        const float dth = vth - mean;
        mean_out = mean;
        sd_out = sqrtf((sq + nt * dth * dth) / (float)top);
    }
    if (lane == 0) { mu[row] = mean_out; sigma[row] = sd_out; }
}

}  // namespace

hipError_t launch_l2norm(float* E, int64_t N, int D, hipStream_t stream) {
    if (N <= 0) return hipSuccess;
    hipLaunchKernelGGL(l2norm_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, stream, E, N, D);
    return hipGetLastError();
}

hipError_t launch_score_pairs(const float* E, int D, const int32_t* ia, const int32_t* ib, int64_t P, float* out, hipStream_t stream) {
    if (P <= 0) return hipSuccess;
    hipLaunchKernelGGL(pair_kernel<0>, dim3((unsigned)((P + 15) / 16)), dim3(256), 0, stream, E, D, ia, ib, P, nullptr, nullptr, out);
    return hipGetLastError();
}

hipError_t launch_asnorm_pairs(const float* E, int D, const float* mu, const float* sigma, const int32_t* ia,
                               const int32_t* ib, int64_t P, float* out, hipStream_t stream) {
    if (P <= 0) return hipSuccess;
    hipLaunchKernelGGL(pair_kernel<1>, dim3((unsigned)((P + 15) / 16)), dim3(256), 0, stream, E, D, ia, ib, P, mu, sigma, out);
    return hipGetLastError();
}

hipError_t launch_trial_crops(int mode, float pexp, const float* F, int n_crops, int D, const int32_t* ia, const int32_t* ib, int64_t P, float* out,
                              hipStream_t stream) {
    if (P <= 0) return hipSuccess;
    if (n_crops <= 0 || D <= 0) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((P + 15) / 16)), block(256);
    if (mode == 0) hipLaunchKernelGGL(trial_crops_kernel<0>, grid, block, 0, stream, F, n_crops, D, ia, ib, P, out, 2.0f);
    else if (mode == 1 && pexp == 2.0f) hipLaunchKernelGGL(trial_crops_kernel<1>, grid, block, 0, stream, F, n_crops, D, ia, ib, P, out, 2.0f);
    else if (mode == 1) hipLaunchKernelGGL(trial_crops_kernel<3>, grid, block, 0, stream, F, n_crops, D, ia, ib, P, out, pexp);
    else if (mode == 2) hipLaunchKernelGGL(trial_crops_kernel<2>, grid, block, 0, stream, F, n_crops, D, ia, ib, P, out, 2.0f);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_mean_crops(const float* F, int64_t n_files, int n_crops, int D, float* out, hipStream_t stream) {
    if (n_files <= 0) return hipSuccess;
    if (n_crops <= 0 || D <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mean_crops_kernel, dim3((unsigned)((n_files * D + 255) / 256)), dim3(256), 0, stream, F, n_files, n_crops, D, out);
    return hipGetLastError();
}

hipError_t launch_topk_stats(const float* S, int64_t rows, int K, int ld, int top, float* mu, float* sigma, hipStream_t stream) {
    if (rows <= 0) return hipSuccess;
    if (top <= 0 || top > K || ld < K) return hipErrorInvalidValue;
    if (ld % 4 == 0 && (reinterpret_cast<uintptr_t>(S) & 15) == 0 && K <= 256 * 24) {      // register-resident rows
        const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
        if (K <= 256 * 8) hipLaunchKernelGGL(topk_stats_reg_kernel<8>, grid, block, 0, stream, S, rows, K, ld, top, mu, sigma);
        else hipLaunchKernelGGL(topk_stats_reg_kernel<24>, grid, block, 0, stream, S, rows, K, ld, top, mu, sigma);
        return hipGetLastError();
    }
    const int Kp = (K + 3) & ~3;
    const size_t row_bytes = (size_t)(Kp + TOPK_CAP) * sizeof(uint32_t);
    int rpw = (int)((150 * 1024) / row_bytes);
    if (rpw < 1) return hipErrorInvalidValue;          // K > 38400: not supported by the LDS-resident selection
    if (rpw > 4) rpw = 4;
    const size_t lds = row_bytes * rpw;
    static DeviceOnce attr;
    if (hipError_t e = set_max_dynamic_lds(attr, reinterpret_cast<const void*>(topk_stats_kernel), 160 * 1024)) return e;
    hipLaunchKernelGGL(topk_stats_kernel, dim3((unsigned)((rows + rpw - 1) / rpw)), dim3(256), lds, stream, S, rows, K, ld, Kp, top, rpw, mu, sigma);
    return hipGetLastError();
}

}  // namespace svhip
