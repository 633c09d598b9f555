// metrics.hip — verification metrics over a scored trial list: the O(P log P) part of the reference's evaluation tail.
//
// Reference: src/utils.py:74-121 tuneThresholdfromScore (sklearn roc_curve / precision_recall_curve, both built on
// _binary_clf_curve: sort the scores, cumulate the labels, keep one point per DISTINCT score) and src/utils.py:221-275
// ComputeErrorRates / ComputeMinDcf (a Python list sort plus two Python loops over every trial).  On 1.2 M trials that is
// seconds of host time per evaluation; here it is one stable radix sort (rocPRIM), two scans and a few streaming kernels.
//
//   sorted ascending (stable: ties keep list order, as Python's sorted() does)      s[0..P), l[0..P)
//   cpos[i] = positives among s[0..i]  (inclusive scan)                             cneg[i] = i + 1 - cpos[i]
//   ComputeErrorRates:  fnrs[i] = cpos[i] / P_pos,  fprs[i] = 1 - cneg[i] / P_neg,  thresholds[i] = s[i]        (float64)
//   ComputeMinDcf:      first i minimising (c_miss * fnrs[i]) * p_target + (c_fa * fprs[i]) * (1 - p_target)
//   _binary_clf_curve:  for the k-th highest distinct value v (its run starts at g in the ascending order):
//                       thr[k] = v, tps[k] = P_pos - cpos[g-1], fps[k] = (P - g) - tps[k]
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <cfloat>
#include "common.h"
#include "kernels.h"

namespace svhip {

namespace {

__global__ __launch_bounds__(256) void nan_to_num_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v = in[i];
    if (v != v) v = 0.0f;                                   // np.nan_to_num: nan -> 0, +-inf -> +-largest finite
    else if (v > FLT_MAX) v = FLT_MAX;
    else if (v < -FLT_MAX) v = -FLT_MAX;
    out[i] = v;
}

// run starts of equal scores in the ascending order: flag[i] = 1 when s[i] != s[i-1] (flag[0] = 1)
__global__ __launch_bounds__(256) void run_flags_kernel(const float* __restrict__ s, int32_t* __restrict__ flag, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    flag[i] = (i == 0 || s[i] != s[i - 1]) ? 1 : 0;
}

// rank[i] = inclusive scan of flag: the run that starts at i is the (rank[i] - 1)-th lowest distinct value
__global__ __launch_bounds__(256) void roc_scatter_kernel(const float* __restrict__ s, const int32_t* __restrict__ flag,
                                                          const int32_t* __restrict__ rank, const int32_t* __restrict__ cpos,
                                                          int64_t n, float* __restrict__ thr, int64_t* __restrict__ fps,
                                                          int64_t* __restrict__ tps) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !flag[i]) return;
    const int64_t nrun = rank[n - 1];
    const int64_t k = nrun - rank[i];                        // 0 = highest distinct value
    const int64_t ppos = cpos[n - 1];
    const int64_t t = ppos - (i > 0 ? cpos[i - 1] : 0);
    thr[k] = s[i];
    tps[k] = t;
    fps[k] = (n - i) - t;
}

#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void error_rates_kernel(const int32_t* __restrict__ cpos, int64_t n, double* __restrict__ fnrs,
                                                          double* __restrict__ fprs) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double ppos = (double)cpos[n - 1], pneg = (double)(n - cpos[n - 1]);
    fnrs[i] = (double)cpos[i] / ppos;                        // x / float(fnrs_norm)          (utils.py:249)
    fprs[i] = 1.0 - (double)(i + 1 - cpos[i]) / pneg;        // 1 - x / float(fprs_norm)      (utils.py:254)
}

struct DcfBest { double c; int64_t i; };

// first minimum of the detection cost over all thresholds (strict '<' in the reference loop, utils.py:268-272)
__global__ __launch_bounds__(256) void min_dcf_kernel(const int32_t* __restrict__ cpos, int64_t n, double p_target, double c_miss,
                                                      double c_fa, DcfBest* __restrict__ part) {
    const double ppos = (double)cpos[n - 1], pneg = (double)(n - cpos[n - 1]);
    const double q = 1.0 - p_target;
    DcfBest best{1.0 / 0.0, INT64_MAX};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double fnr = (double)cpos[i] / ppos;
        const double fpr = 1.0 - (double)(i + 1 - cpos[i]) / pneg;
        const double c = c_miss * fnr * p_target + c_fa * fpr * q;           // left to right, no contraction: Python's order
        if (c < best.c || (c == best.c && i < best.i)) best = DcfBest{c, i};
    }
    __shared__ DcfBest sh[256];
    sh[threadIdx.x] = best;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const DcfBest b = sh[threadIdx.x + o];
            DcfBest& a = sh[threadIdx.x];
            if (b.c < a.c || (b.c == a.c && b.i < a.i)) a = b;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}

__global__ void min_dcf_final_kernel(const DcfBest* __restrict__ part, int nparts, const float* __restrict__ s, double p_target,
                                     double c_miss, double c_fa, double* __restrict__ out_dcf, float* __restrict__ out_thr) {
    if (threadIdx.x || blockIdx.x) return;
    DcfBest a = part[0];
    for (int j = 1; j < nparts; ++j) { const DcfBest b = part[j]; if (b.c < a.c || (b.c == a.c && b.i < a.i)) a = b; }
    const double m = c_miss * p_target, f = c_fa * (1.0 - p_target);
    *out_dcf = a.c / (m < f ? m : f);                        // c_def = min(...)            (utils.py:274-275)
    *out_thr = s[a.i];
}

inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

// workspace carve-up for P trials: s_in | s_sorted | l_sorted | cpos | flag | rank | DcfBest parts | rocPRIM temp
size_t metrics_workspace_bytes(int64_t P) {
    size_t t_sort = 0, t_scan = 0;
    (void)rocprim::radix_sort_pairs<rocprim::default_config, const float*, float*, const int32_t*, int32_t*>(
        nullptr, t_sort, nullptr, nullptr, nullptr, nullptr, (size_t)P, 0, 32, nullptr);
    (void)rocprim::inclusive_scan<rocprim::default_config, const int32_t*, int32_t*, rocprim::plus<int32_t>>(
        nullptr, t_scan, nullptr, nullptr, (size_t)P, rocprim::plus<int32_t>(), nullptr);
    const size_t tmp = t_sort > t_scan ? t_sort : t_scan;
    return 6 * align_up((size_t)P * 4) + align_up(1024 * sizeof(DcfBest)) + align_up(tmp) + 256;
}

namespace {
struct Ws {
    float *s_in, *s; int32_t *l, *cpos, *flag, *rank; DcfBest* parts; void* tmp; size_t tmp_bytes;
    Ws(void* base, int64_t P, size_t total) {
        char* p = static_cast<char*>(base);
        const size_t a = align_up((size_t)P * 4);
        s_in = (float*)p; p += a; s = (float*)p; p += a; l = (int32_t*)p; p += a; cpos = (int32_t*)p; p += a;
        flag = (int32_t*)p; p += a; rank = (int32_t*)p; p += a; parts = (DcfBest*)p; p += align_up(1024 * sizeof(DcfBest));
        tmp = p; tmp_bytes = total - (size_t)(p - static_cast<char*>(base));
    }
};
inline int blocks(int64_t n) { return (int)((n + 255) / 256); }
}  // namespace

// stable ascending sort of (score, label) + inclusive positive counts
hipError_t metrics_sort_scan(const float* scores, const int32_t* labels, int64_t P, bool nan_to_num, void* ws_base, size_t ws_bytes,
                             hipStream_t st) {
    if (P <= 0 || P >= (int64_t)1 << 31) return hipErrorInvalidValue;
    Ws w(ws_base, P, ws_bytes);
    const float* keys = scores;
    if (nan_to_num) {
        hipLaunchKernelGGL(nan_to_num_kernel, dim3(blocks(P)), dim3(256), 0, st, scores, w.s_in, P);
        keys = w.s_in;
    }
    size_t tb = w.tmp_bytes;
    hipError_t e = rocprim::radix_sort_pairs(w.tmp, tb, keys, w.s, labels, w.l, (size_t)P, 0, 32, st);
    if (e != hipSuccess) return e;
    tb = w.tmp_bytes;
    e = rocprim::inclusive_scan(w.tmp, tb, (const int32_t*)w.l, w.cpos, (size_t)P, rocprim::plus<int32_t>(), st);
    if (e != hipSuccess) return e;
    return hipGetLastError();
}

hipError_t metrics_roc_points(int64_t P, void* ws_base, size_t ws_bytes, float* thr, int64_t* fps, int64_t* tps, int32_t** n_runs_dev,
                              hipStream_t st) {
    Ws w(ws_base, P, ws_bytes);
    hipLaunchKernelGGL(run_flags_kernel, dim3(blocks(P)), dim3(256), 0, st, (const float*)w.s, w.flag, P);
    size_t tb = w.tmp_bytes;
    hipError_t e = rocprim::inclusive_scan(w.tmp, tb, (const int32_t*)w.flag, w.rank, (size_t)P, rocprim::plus<int32_t>(), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(roc_scatter_kernel, dim3(blocks(P)), dim3(256), 0, st, (const float*)w.s, (const int32_t*)w.flag,
                       (const int32_t*)w.rank, (const int32_t*)w.cpos, P, thr, fps, tps);
    *n_runs_dev = w.rank + (P - 1);
    return hipGetLastError();
}

hipError_t metrics_error_rates(int64_t P, void* ws_base, size_t ws_bytes, double* fnrs, double* fprs, float* thresholds, hipStream_t st) {
    Ws w(ws_base, P, ws_bytes);
    hipLaunchKernelGGL(error_rates_kernel, dim3(blocks(P)), dim3(256), 0, st, (const int32_t*)w.cpos, P, fnrs, fprs);
    hipError_t e = hipMemcpyAsync(thresholds, w.s, (size_t)P * 4, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return e;
    return hipGetLastError();
}

hipError_t metrics_min_dcf(int64_t P, void* ws_base, size_t ws_bytes, double p_target, double c_miss, double c_fa, double* dcf_dev,
                           float* thr_dev, hipStream_t st) {
    Ws w(ws_base, P, ws_bytes);
    const int nb = (int)(P < 256 * 1024 ? (P + 255) / 256 : 1024);
    hipLaunchKernelGGL(min_dcf_kernel, dim3(nb), dim3(256), 0, st, (const int32_t*)w.cpos, P, p_target, c_miss, c_fa, w.parts);
    hipLaunchKernelGGL(min_dcf_final_kernel, dim3(1), dim3(64), 0, st, (const DcfBest*)w.parts, nb, (const float*)w.s, p_target, c_miss,
                       c_fa, dcf_dev, thr_dev);
    return hipGetLastError();
}

}  // namespace svhip
