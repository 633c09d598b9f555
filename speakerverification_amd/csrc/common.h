// common.h — shared device/host helpers for libsvhip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace svhip {

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int WAVE = 64;

enum Act : int { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2, ACT_TANH = 3, ACT_SIGMOID = 4, ACT_LRELU03 = 5, ACT_LRELU001 = 6 };
enum Pad : int { PAD_REFLECT = 0, PAD_ZERO = 1 };

__device__ __forceinline__ float bf2f(bf16_t v) { return static_cast<float>(v); }
__device__ __forceinline__ bf16_t f2bf(float v) { return static_cast<bf16_t>(v); }

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return static_cast<float>(v); }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return static_cast<bf16_t>(v); }

__device__ __forceinline__ float apply_act(float x, int act) {
    switch (act) {
        case ACT_RELU: return fmaxf(x, 0.0f);
        case ACT_GELU: return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));   // nn.GELU() exact form
        case ACT_TANH: return tanhf(x);
        case ACT_SIGMOID: return 1.0f / (1.0f + expf(-x));
        case ACT_LRELU03: return x > 0.0f ? x : 0.3f * x;
        case ACT_LRELU001: return x > 0.0f ? x : 0.01f * x;
        default: return x;
    }
}

// reflect index into [0, n): -1 -> 1, n -> n-2 (torch 'reflect' padding; |overshoot| < n)
__device__ __forceinline__ int reflect_idx(int t, int n) {
    t = t < 0 ? -t : t;
    return t >= n ? 2 * (n - 1) - t : t;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// 16-byte vector of activations: 4 floats or 8 bf16.
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int N = 4;
    f32x4 v;
    __device__ __forceinline__ float get(int i) const { return v[i]; }
    __device__ __forceinline__ void set(int i, float x) { v[i] = x; }
};
template <> struct Vec16<bf16_t> {
    static constexpr int N = 8;
    bf16x8 v;
    __device__ __forceinline__ float get(int i) const { return static_cast<float>(v[i]); }
    __device__ __forceinline__ void set(int i, float x) { v[i] = static_cast<bf16_t>(x); }
};

}  // namespace svhip
