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

__device__ __forceinline__ uint32_t fkey(float v) {       // order-preserving float -> uint map
    const uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
__device__ __forceinline__ int wave_isum(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// One wavefront per row; the row (K fp32 scores) is staged in LDS as sortable keys, the `top`-th
// largest key is found by a 32-step bitwise search (count(key >= candidate) by ds_read_b128 sweeps),
// then mean / population std of the selected values (ties at the threshold counted exactly).
__global__ __launch_bounds__(256) void topk_stats_kernel(const float* __restrict__ S, int64_t rows, int K, int Kp, int top,
                                                         int rows_per_wg, float* __restrict__ mu, float* __restrict__ sigma) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t* keys = reinterpret_cast<uint32_t*>(smem) + (size_t)wave * Kp;
    const int64_t row = (int64_t)blockIdx.x * rows_per_wg + wave;
    const bool active = (wave < rows_per_wg) && (row < rows);
    if (active) {
        const float* __restrict__ s = S + row * K;
        for (int k = lane; k < Kp; k += 64) keys[k] = (k < K) ? fkey(s[k]) : 0u;    // pad = smallest key
    }
    __syncthreads();
    if (!active) return;
    const u32x4* k4 = reinterpret_cast<const u32x4*>(keys);
    const int n4 = Kp >> 2;
    uint32_t prefix = 0;
    for (int bit = 31; bit >= 0; --bit) {
        const uint32_t cand = prefix | (1u << bit);
        int cnt = 0;
        for (int i = lane; i < n4; i += 64) {
            const u32x4 v = k4[i];
            cnt += (v[0] >= cand) + (v[1] >= cand) + (v[2] >= cand) + (v[3] >= cand);
        }
        cnt = wave_isum(cnt);
        if (cnt >= top) prefix = cand;
    }
    // prefix == key of the top-th largest element
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
    const float var = (sq + nt * dth * dth) / (float)top;
    if (lane == 0) { mu[row] = mean; sigma[row] = sqrtf(var); }
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

hipError_t launch_topk_stats(const float* S, int64_t rows, int K, int top, float* mu, float* sigma, hipStream_t stream) {
    if (rows <= 0) return hipSuccess;
    if (top <= 0 || top > K) return hipErrorInvalidValue;
    const int Kp = (K + 3) & ~3;
    const size_t row_bytes = (size_t)Kp * sizeof(uint32_t);
    int rpw = (int)((150 * 1024) / row_bytes);
    if (rpw < 1) return hipErrorInvalidValue;          // K > 38400: not supported by the LDS-resident selection
    if (rpw > 4) rpw = 4;
    const size_t lds = row_bytes * rpw;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(topk_stats_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(topk_stats_kernel, dim3((unsigned)((rows + rpw - 1) / rpw)), dim3(256), lds, stream, S, rows, K, Kp, top, rpw, mu, sigma);
    return hipGetLastError();
}

}  // namespace svhip
