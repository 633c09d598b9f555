// common.h — shared device/host helpers for libsvhip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace svhip {

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// fp16 (IEEE half): the second 16-bit storage type.  RawNet2's 16-bit path runs on it (SVHIP_F16): same MFMA rate as bf16, three more
// mantissa bits — on RawNet2 the bf16 rounding of the conv WEIGHTS alone moves an embedding by up to 9 % of its scale (cosine 0.994
// to the fp32 reference; tests/analysis/rn_bf16_sites.py), fp16 storage keeps it within 1 % (cosine >= 0.9999)
typedef _Float16 f16_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int WAVE = 64;

// storage / MFMA operand type of a launch: the launchers' `dt` argument (a plain `bool bf16` converts to DT_F32 / DT_BF16)
enum DType : int { DT_F32 = 0, DT_BF16 = 1, DT_F16 = 2 };

enum Act : int { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2, ACT_TANH = 3, ACT_SIGMOID = 4, ACT_LRELU03 = 5, ACT_LRELU001 = 6 };
enum Pad : int { PAD_REFLECT = 0, PAD_ZERO = 1 };

__device__ __forceinline__ float bf2f(bf16_t v) { return static_cast<float>(v); }
__device__ __forceinline__ bf16_t f2bf(float v) { return static_cast<bf16_t>(v); }

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return static_cast<float>(v); }
template <> __device__ __forceinline__ float to_f32<f16_t>(f16_t v) { return static_cast<float>(v); }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ f16_t from_f32<f16_t>(float v) { return static_cast<f16_t>(v); }
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
// cache-policy bits of a global_load_lds (the builtin's last argument; gfx940+: sc0 = 1, nt = 2, sc1 = 16).  NT marks a stream that is read
// once: its lines are the first to leave L2 / the MALL, so they do not displace the tensors the next kernel is about to read
// (round 4: se_apply 145 -> 103 us with non-temporal loads of its two read-once inputs).
constexpr int CPOL_NT = 2;

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

template <> struct Vec16<f16_t> {
    static constexpr int N = 8;
    f16x8 v;
    __device__ __forceinline__ float get(int i) const { return static_cast<float>(v[i]); }
    __device__ __forceinline__ void set(int i, float x) { v[i] = static_cast<f16_t>(x); }
};

// 16-byte non-temporal load of a read-once stream (see CPOL_NT)
template <typename T> __device__ __forceinline__ Vec16<T> ld_nt(const T* p) {
    Vec16<T> r;
    r.v = __builtin_nontemporal_load(reinterpret_cast<const decltype(r.v)*>(p));
    return r;
}

// Per-type pieces of the 16-bit kernels: H = bf16_t or f16_t.  Fragments travel as raw 16-byte vectors (bf16x8 in the kernels'
// declarations whatever H is); only the MFMA opcode and the fp32 <-> 16-bit conversions differ.
template <typename H> struct Half16;
template <> struct Half16<bf16_t> {
    static constexpr bool F16 = false;
    static __device__ __forceinline__ f32x4 mfma16(const bf16x8& a, const bf16x8& b, const f32x4& c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ f32x16 mfma32(const bf16x8& a, const bf16x8& b, const f32x16& c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
    // two 16-bit values in one dword <-> fp32 (unpack exact, pack round-to-nearest-even: v_cvt_pk_bf16_f32)
    static __device__ __forceinline__ float lo(uint32_t w) { return __uint_as_float(w << 16); }
    static __device__ __forceinline__ float hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
    static __device__ __forceinline__ uint32_t pack2(float a, float b) {
        typedef float f32x2_ __attribute__((ext_vector_type(2)));
        typedef bf16_t bf16x2_ __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_{a, b}, bf16x2_));
    }
};
template <> struct Half16<f16_t> {
    static constexpr bool F16 = true;
    static __device__ __forceinline__ f32x4 mfma16(const bf16x8& a, const bf16x8& b, const f32x4& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x16 mfma32(const bf16x8& a, const bf16x8& b, const f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    typedef f16_t f16x2_ __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ float lo(uint32_t w) { return static_cast<float>(__builtin_bit_cast(f16x2_, w)[0]); }
    static __device__ __forceinline__ float hi(uint32_t w) { return static_cast<float>(__builtin_bit_cast(f16x2_, w)[1]); }
    static __device__ __forceinline__ uint32_t pack2(float a, float b) {      // v_cvt_pk_f16_f32 (round to nearest even; overflow -> inf)
        typedef float f32x2_ __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_{a, b}, f16x2_));
    }
};

// ---- the split ("X3") forms: fp32 values as hi | lo planes of a 16-bit type, a product as three MFMAs (hi.hi + hi.lo + lo.hi) ----------
// Round 4: the planes are IEEE half, not bf16.  bf16 planes carry 8 + 8 significant bits (v - hi - lo ~ 2^-17 |v|: the error of rounds
// 2 - 3's f32x3 mode); fp16 planes carry 11 + 11 (2^-23 |v| while lo is a normal half, i.e. |v| >= 2^-3; below that lo is a subnormal
// and the residual is at most 2^-25 ABSOLUTE) at the same MFMA rate: the mode becomes fp32-grade.  -DSVHIP_X3_BF16 rebuilds
// the old planes (A/B).
// RANGE (round 5): the conversion is the plain one — |v| > 65504 becomes inf, a NaN stays a NaN — so that what the planes cannot carry
// shows up in the embeddings, where every forward looks (emb_out_kernel -> SVHIP_ERR_NONFINITE).  Round 4 clamped with v_med3_f32, which
// kept the result finite and WRONG, and turned a NaN input into +-65504 (a NaN sample gave a finite, meaningless embedding on f32x3
// handles only: tests/test_gpu_rawnet2.py::test_nonfinite_input_is_reported_on_every_handle_kind).
#ifdef SVHIP_X3_BF16
typedef bf16_t x3_t;
#else
typedef f16_t x3_t;
#endif
typedef Half16<x3_t> X3H;
typedef x3_t x3x2_t __attribute__((ext_vector_type(2)));
typedef x3_t x3x4_t __attribute__((ext_vector_type(4)));
typedef x3_t x3x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ x3_t x3_hi(float v) { return static_cast<x3_t>(v); }
__device__ __forceinline__ x3_t x3_lo(float v, x3_t h) { return x3_hi(v - static_cast<float>(h)); }
// two values -> one dword of their hi parts and one of their lo parts
__device__ __forceinline__ void x3_split2(float a, float b, uint32_t& hd, uint32_t& ld) {
    const x3_t ha = x3_hi(a), hb = x3_hi(b);
    hd = __builtin_bit_cast(uint32_t, x3x2_t{ha, hb});
    ld = __builtin_bit_cast(uint32_t, x3x2_t{x3_lo(a, ha), x3_lo(b, hb)});
}

// ---- exact power-of-two operand scaling of the half-plane forms (asnorm_fused.hip "operand scaling"; elementwise.hip / gemm_pw3.hip: the
// network input of F32X3 handles, round 6) ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float pow2_scale_of_bits(uint32_t bits) {      // s = 2^k with (max |x|) * s in [64, 128)
    const int e = (int)((bits >> 23) & 255u);
    if (e == 0) return 1.0f;                                              // zero (or subnormal) row: nothing to scale
    const int se = min(253, max(1, 260 - e));                             // biased exponent of s = 127 + 6 - (e - 127)
    return __uint_as_float((uint32_t)se << 23);
}
// |x| as ordered bits for the max-|x| searches; inf / NaN count as 0: a non-finite element must not set the scale of the
// FINITE elements around it (exponent 255 would give 2^-122 and flush a whole matrix to zero) — it stays inf / NaN in its own row or column
__device__ __forceinline__ uint32_t finite_abs_bits(uint32_t w) { const uint32_t a = w & 0x7fffffffu; return a < 0x7f800000u ? a : 0u; }
__device__ __forceinline__ float pow2_inverse(float s) { return __uint_as_float((254u - (__float_as_uint(s) >> 23)) << 23); }      // s = 2^k, biased k in [1, 253]


}  // namespace svhip
