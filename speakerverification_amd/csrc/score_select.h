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
// `nq` (wave-uniform, <= PER): only the first nq keys of a lane can be populated (a compacted list: key j of lane l is element l + 64 j);
// `bounded`: the caller guarantees at least `top` populated (non-zero) keys — the search then starts below the bits that the smallest and
// the largest populated key have in common (candidate scores above one threshold share sign, most of the exponent: 8 - 12 of the 32 steps).
template <int PER>
__device__ __forceinline__ void select_stats(const uint32_t (&k)[PER], int top, float& mean_out, float& sd_out, int nq = PER, bool bounded = false) {
    // (the count of a step is the sum of the population counts of PER lane masks: scalar work, no cross-lane shuffle — as
    //  per-lane counts + a six-step butterfly the search was 192 dependent ds_bpermute round trips per embedding and the candidate
    //  kernel took as long as the matrix kernel it runs beside)
    uint32_t prefix = 0;
    int first_bit = 31;
    if (bounded) {
        uint32_t kmax = 0, kmin = 0xffffffffu;
#pragma unroll
        for (int j = 0; j < PER; ++j)
            if (j < nq) { kmax = max(kmax, k[j]); kmin = min(kmin, k[j] ? k[j] : 0xffffffffu); }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, off, 64));
            kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, off, 64));
        }
        const uint32_t diff = __builtin_amdgcn_readfirstlane(kmax ^ kmin);
        if (diff == 0) { first_bit = -1; prefix = __builtin_amdgcn_readfirstlane(kmax); }          // every populated key is the same
        else {
            first_bit = 31 - __builtin_clz(diff);                                                   // highest bit in which two populated keys differ
            prefix = __builtin_amdgcn_readfirstlane(kmax) & ~((2u << first_bit) - 1u);            // (first_bit = 31: the mask is 0)
        }
    }
    for (int bit = first_bit; bit >= 0; --bit) {
        const uint32_t cand = prefix | (1u << bit);
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j)
            if (j < nq) cnt += __popcll(__ballot(k[j] >= cand));
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
