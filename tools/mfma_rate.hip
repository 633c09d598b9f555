// developer probe (round 4): FLOP/s of bare MFMA loops on RANDOM data, 32x32x16 against 16x16x32 (bf16 and f16), one and two waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o tools/_mfma_rate.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE, bool F16>   // SHAPE 0: 32x32x16, 1: 16x16x32.  Same output tile per wave: 64 x 64 (4 accumulators of 32x32 / 16 of 16x16)
__global__ __launch_bounds__(512) void k(const uint4* __restrict__ src, float* out, int iters) {
    extern __shared__ char lds[];
    const int tid = threadIdx.x;
    uint4 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = src[(blockIdx.x * 512 + tid) * 8 + i]; b[i] = src[(blockIdx.x * 512 + tid) * 8 + 4 + i]; }
    f32x16 acc32[4]; f32x4 acc16[16];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc32[i][r] = 0.f;
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) acc16[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (SHAPE == 0) {
            // 64 x 64 x 32: 2 x 2 tiles x 2 k-steps = 8 MFMAs of 32768 FLOP
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if (F16) acc32[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i + 2 * ks]), __builtin_bit_cast(f16x8, b[j + 2 * ks]), acc32[i * 2 + j], 0, 0, 0);
                        else acc32[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i + 2 * ks]), __builtin_bit_cast(bf16x8, b[j + 2 * ks]), acc32[i * 2 + j], 0, 0, 0);
                    }
        } else {
            // 64 x 64 x 32: 4 x 4 tiles x 1 k-step = 16 MFMAs of 16384 FLOP
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (F16) acc16[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[i]), __builtin_bit_cast(f16x8, b[j]), acc16[i * 4 + j], 0, 0, 0);
                    else acc16[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), acc16[i * 4 + j], 0, 0, 0);
                }
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc32[i][r];
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) s += acc16[i][r];
    out[blockIdx.x * 512 + tid] = s;
}

template <int SHAPE, bool F16> void run(const char* name, int threads, const uint4* src, float* out) {
    const int iters = 40000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<SHAPE, F16>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<SHAPE, F16><<<256, threads, 100 * 1024>>>(src, out, iters); hipDeviceSynchronize();
    hipEventRecord(e0);
    k<SHAPE, F16><<<256, threads, 100 * 1024>>>(src, out, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = 256.0 * (threads / 64) * (double)iters * 2.0 * 64 * 64 * 32;
    printf("%-18s waves/SIMD %d: %7.1f TFLOP/s (%.2f ms)\n", name, threads / 256, flop / (ms * 1e-3) / 1e12, ms);
}


template <int SHAPE>   // fp32 MFMA: 0: 32x32x2, 1: 16x16x4; output tile per wave 64 x 64, k = 8 per iteration
__global__ __launch_bounds__(512) void kf(const uint4* __restrict__ src, float* out, int iters) {
    extern __shared__ char lds[];
    const int tid = threadIdx.x;
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = 1.0f + 1e-3f * (float)((src[(blockIdx.x * 512 + tid) * 8 + i].x) & 1023); b[i] = 1.0f - 1e-3f * (float)((src[(blockIdx.x * 512 + tid) * 8 + i].y) & 1023); }
    f32x16 acc32[4]; f32x4 acc16[16];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc32[i][r] = 0.f;
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) acc16[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (SHAPE == 0) {
            // 64 x 64 x 8: 2 x 2 tiles x 4 k-steps of 2 = 16 MFMAs of 4096 FLOP
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc32[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i + 2 * ks], b[j + 2 * ks], acc32[i * 2 + j], 0, 0, 0);
        } else {
            // 64 x 64 x 8: 4 x 4 tiles x 2 k-steps of 4 = 32 MFMAs of 2048 FLOP
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc16[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i + 4 * ks], b[j + 4 * ks], acc16[i * 4 + j], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc32[i][r];
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) s += acc16[i][r];
    out[blockIdx.x * 512 + tid] = s;
}
template <int SHAPE> void runf(const char* name, int threads, const uint4* src, float* out) {
    const int iters = 20000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kf<SHAPE>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kf<SHAPE><<<256, threads, 100 * 1024>>>(src, out, iters); hipDeviceSynchronize();
    hipEventRecord(e0);
    kf<SHAPE><<<256, threads, 100 * 1024>>>(src, out, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = 256.0 * (threads / 64) * (double)iters * 2.0 * 64 * 64 * 8;
    printf("%-18s waves/SIMD %d: %7.1f TFLOP/s (%.2f ms)\n", name, threads / 256, flop / (ms * 1e-3) / 1e12, ms);
}

int main() {
    const size_t n = 256 * 512 * 8;
    uint4* h = (uint4*)malloc(n * 16);
    // random bf16 / f16 bit patterns of moderate magnitude: sign random, exponent near 1.0, mantissa random
    unsigned short* hs = (unsigned short*)h;
    srand(1);
    for (size_t i = 0; i < n * 8; ++i) hs[i] = (unsigned short)(((rand() & 1) << 15) | (0x3c00 + (rand() & 0x3ff)) - ((rand() & 3) << 10));
    uint4* src; float* out;
    hipMalloc(&src, n * 16); hipMalloc(&out, 256 * 512 * 4);
    hipMemcpy(src, h, n * 16, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<0, false>("32x32x16 bf16", 256, src, out); run<1, false>("16x16x32 bf16", 256, src, out);
        run<0, false>("32x32x16 bf16", 512, src, out); run<1, false>("16x16x32 bf16", 512, src, out);
        run<0, true>("32x32x16 f16", 256, src, out); run<1, true>("16x16x32 f16", 256, src, out);
        run<0, true>("32x32x16 f16", 512, src, out); run<1, true>("16x16x32 f16", 512, src, out);
    }
    for (int rep = 0; rep < 2; ++rep) { runf<0>("32x32x2 f32", 256, src, out); runf<1>("16x16x4 f32", 256, src, out); runf<0>("32x32x2 f32", 512, src, out); runf<1>("16x16x4 f32", 512, src, out); }
    hipMemset(src, 0, n * 16);
    run<0, false>("32x32x16 bf16 ZEROS", 512, src, out); run<1, false>("16x16x32 bf16 ZEROS", 512, src, out);
    return 0;
}
