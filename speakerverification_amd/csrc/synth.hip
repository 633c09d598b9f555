// synth.hip — synthetic waveforms generated on the device from a counter-based RNG ("1 M utterances
// generated on-device per shard from a counter-based RNG (Philox) — do not stream 128 GB of waveforms over PCIe").
//
// Sample s of utterance u is a pure function of (seed, u, s): Philox4x32-10 with key = (seed lo, seed hi) and counter
// = (s / 4, u lo, u hi, 0) yields four 32-bit words = two Box-Muller pairs = samples 4*(s/4) .. +3, scaled to
// 0.1 * N(0, 1) and clipped to [-1, 1] (the amplitude convention of the fixed config-1/2 waveforms).  Any rank can therefore
// produce exactly its block [first_utt, first_utt + B) of the global utterance list; oracle/synthwave.py restates it in numpy.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"
#include "kernels.h"

namespace svhip {

namespace {

__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

__global__ __launch_bounds__(256) void synth_wave_kernel(float* __restrict__ out, uint64_t seed, int64_t first_utt, int B, int L) {
    const int quads = L >> 2;                            // L % 4 == 0 (checked by the launcher)
    const int64_t total = (int64_t)B * quads;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / quads), q = (int)(i - (int64_t)b * quads);
        const uint64_t u = (uint64_t)(first_utt + b);
        uint32_t c[4] = {(uint32_t)q, (uint32_t)u, (uint32_t)(u >> 32), 0u};
        philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
        float v[4];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const float u1 = ((float)c[2 * p] + 0.5f) * 2.3283064365386963e-10f;        // (0, 1]: 2^-32 scaling of a 32-bit word
            const float u2 = ((float)c[2 * p + 1] + 0.5f) * 2.3283064365386963e-10f;
            const float rad = sqrtf(-2.0f * logf(u1));
            float sn, cs;
            sincosf(6.283185307179586f * u2, &sn, &cs);
            v[2 * p] = rad * cs; v[2 * p + 1] = rad * sn;
        }
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fminf(fmaxf(0.1f * v[e], -1.0f), 1.0f);
        *reinterpret_cast<f32x4*>(out + (int64_t)b * L + 4 * q) = o;
    }
}

}  // namespace

hipError_t launch_synth_wave(float* out, uint64_t seed, int64_t first_utt, int B, int L, hipStream_t stream) {
    if (!out || B <= 0 || L <= 0 || (L & 3)) return hipErrorInvalidValue;
    const int64_t total = (int64_t)B * (L >> 2);
    const int blocks = (int)((total + 255) / 256 < 256 * 64 ? (total + 255) / 256 : 256 * 64);
    hipLaunchKernelGGL(synth_wave_kernel, dim3(blocks), dim3(256), 0, stream, out, seed, first_utt, B, L);
    return hipGetLastError();
}

}  // namespace svhip
