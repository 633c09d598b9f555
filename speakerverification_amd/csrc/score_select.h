// score_select.h — order-preserving float keys and the exact top-`top` statistics of a wave-held key set
// (shared by score.hip's slab kernels and asnorm_fused.hip's candidate kernel).
#pragma once
#include "common.h"

namespace svhip {

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

// Exact mean / population std of the `top` largest of n sortable keys held as `per` keys per lane
// (unused slots = key 0, the smallest): 32-step bitwise search for the top-th largest key, ties at the
// threshold counted exactly.
template <int PER>
__device__ __forceinline__ void select_stats(const uint32_t (&k)[PER], int top, float& mean_out, float& sd_out) {
    // (the count of a step is the sum of the population counts of PER lane masks: scalar work, no cross-lane shuffle — as
    //  per-lane counts + a six-step butterfly the search was 192 dependent ds_bpermute round trips per embedding and the candidate
    //  kernel took as long as the matrix kernel it runs beside)
    uint32_t prefix = 0;
    for (int bit = 31; bit >= 0; --bit) {
        const uint32_t cand = prefix | (1u << bit);
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) cnt += __popcll(__ballot(k[j] >= cand));
        if (cnt >= top) prefix = cand;
    }
    const float vth = fkey_inv(prefix);
    float sum = 0.0f;
    int cgt = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j)
        if (k[j] > prefix) { sum += fkey_inv(k[j]); ++cgt; }
    sum = wave_sum(sum);
    cgt = wave_isum(cgt);
    const float nt = (float)(top - cgt);
    const float mean = (sum + nt * vth) / (float)top;
    float sq = 0.0f;
#pragma unroll
    for (int j = 0; j < PER; ++j)
        if (k[j] > prefix) { const float d = fkey_inv(k[j]) - mean; sq = fmaf(d, d, sq); }
    sq = wave_sum(sq);
    const float dth = vth - mean;
    mean_out = mean;
    sd_out = sqrtf((sq + nt * dth * dth) / (float)top);
}


}  // namespace svhip
