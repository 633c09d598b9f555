// gemm_epi.h — epilogue arithmetic shared by the bf16 256 x 256 GEMM kernels (gemm_pw2.hip, gemm_pw3.hip).
#pragma once
#include "common.h"

namespace svhip {

enum Epi : int { EPI_NONE = 0, EPI_RELU = 1, EPI_GELU = 2, EPI_LRELU03 = 4, EPI_BN_LRELU03 = 5 };   // 5: affine first, then LeakyReLU(0.3)

// GELU for the bf16 path: x * sigmoid(x * (c0 + c1 s + c2 s^2)), s = min(x^2, 52) (the polynomial peaks at s = 52.6, so the
// clamp keeps it monotone); coefficients are a minimax fit to 0.5 x (1 + erf(x / sqrt 2)) over [-8, 8]: |err| <= 2.6e-5 absolute,
// i.e. below half a bf16 ulp of every |gelu| >= 0.0066 (the output is rounded to bf16 right after).  Two elements at a
// time so the polynomial runs on v_pk_mul_f32 / v_pk_fma_f32; coefficients pre-multiplied by -log2(e) for v_exp_f32.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t gelu_pair(f32x2_t x) {
    constexpr float L2E = 1.44269504088896340736f;
    const f32x2_t k0 = {-1.5950157685710367f * L2E, -1.5950157685710367f * L2E};
    const f32x2_t k1 = {-0.07401129204455145f * L2E, -0.07401129204455145f * L2E};
    const f32x2_t k2 = {0.0007030335770637797f * L2E, 0.0007030335770637797f * L2E};
    f32x2_t s = x * x;
    s.x = fminf(s.x, 52.0f); s.y = fminf(s.y, 52.0f);
    f32x2_t q = s * k2 + k1;
    q = q * s + k0;
    const f32x2_t z = x * q;
    f32x2_t e = {__builtin_amdgcn_exp2f(z.x), __builtin_amdgcn_exp2f(z.y)};
    e = e + f32x2_t{1.0f, 1.0f};
    const f32x2_t r = {__builtin_amdgcn_rcpf(e.x), __builtin_amdgcn_rcpf(e.y)};
    return x * r;
}

// (the accumulators start at the bias, so the activation sees the biased value directly)
template <int EPI>
__device__ __forceinline__ void act4(float (&v)[4], const f32x4& a, const f32x4& sc4, const f32x4& sh4) {
    if (EPI == EPI_GELU) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x2_t x = {a[2 * h], a[2 * h + 1]};
            const f32x2_t g = gelu_pair(x);
            const f32x2_t sc = {sc4[2 * h], sc4[2 * h + 1]}, sh = {sh4[2 * h], sh4[2 * h + 1]};
            const f32x2_t y = g * sc + sh;
            v[2 * h] = y.x; v[2 * h + 1] = y.y;
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = a[e];
            if (EPI == EPI_RELU) t = fmaxf(t, 0.0f);
            if (EPI == EPI_LRELU03) t = fmaxf(t, 0.3f * t);               // == t > 0 ? t : 0.3 t, without the compare mask
            t = fmaf(t, sc4[e], sh4[e]);
            if (EPI == EPI_BN_LRELU03) t = fmaxf(t, 0.3f * t);            // RawNet2: conv -> bn2 -> lrelu (act1 none, act2 lrelu)
            v[e] = t;
        }
    }
}

// two floats -> one dword of two bf16 (round to nearest even, v_cvt_pk_bf16_f32)
__device__ __forceinline__ uint32_t bf16_pack2(float a, float b) {
    typedef bf16_t bf16x2_ __attribute__((ext_vector_type(2)));
    const bf16x2_ r = __builtin_convertvector(f32x2_t{a, b}, bf16x2_);
    return __builtin_bit_cast(uint32_t, r);
}

// sum over the 16 lanes of a DPP row (lanes that share lane >> 4); every lane of the row gets the total
__device__ __forceinline__ float row16_sum(float x) {
#define SVHIP_DPP_ADD(ctrl) x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), (ctrl), 0xf, 0xf, true))
    SVHIP_DPP_ADD(0xB1);      // quad_perm [1, 0, 3, 2]
    SVHIP_DPP_ADD(0x4E);      // quad_perm [2, 3, 0, 1]
    SVHIP_DPP_ADD(0x141);     // row_half_mirror
    SVHIP_DPP_ADD(0x140);     // row_mirror
#undef SVHIP_DPP_ADD
    return x;
}

}  // namespace svhip
