// developer probe (round 4): issue cycles per wave instruction of the vector ops an epilogue is made of (gfx950).
// hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* cyc, int iters, float seed) {
    extern __shared__ char big_lds[];      // 100 KiB requested: one workgroup per CU
    float a[8]; f2 p[8]; h2 h[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; p[i] = f2{a[i], a[i] + 1}; h[i] = h2{(_Float16)(a[i] * 1e-3f), (_Float16)0.5f}; }
    const float c = seed * 0.999f; const f2 pc = {c, c}; const h2 hc = {(_Float16)0.999f, (_Float16)0.999f};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
                if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(pc));
                if (OP == 2) asm volatile("v_pk_fma_f16 %0, %0, %1, %1" : "+v"(h[i]) : "v"(hc));
                if (OP == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                if (OP == 4) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
                if (OP == 5) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %1" : "=v"(h[i]) : "v"(a[i]));
                if (OP == 6) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
                if (OP == 7) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(h[i]) : "v"(hc));
                if (OP == 8) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(a[i]) : "v"(h[i]));
                if (OP == 9) asm volatile("v_pk_min_f16 %0, %0, %1" : "+v"(h[i]) : "v"(hc));
                if (OP == 10) asm volatile("v_exp_f16 %0, %0" : "+v"(h[i]));
                if (OP == 11) asm volatile("v_rcp_f16 %0, %0" : "+v"(h[i]));
                if (OP == 12) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
                if (OP == 13) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y + (float)h[i].x + (float)h[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int OP> void run(const char* name, int threads) {
    float* out; unsigned long long* cyc; hipMalloc(&out, 4 * 1024 * 256); hipMalloc(&cyc, 8);
    const int iters = 20000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<256, threads, 100 * 1024>>>(out, cyc, iters, 1.0f); hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<256, threads, 100 * 1024>>>(out, cyc, iters, 1.0f);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double per_simd = (double)(threads / 256) * iters * 32.0;       // wave instructions one SIMD executed
    printf("%-22s waves/SIMD %d: %6.3f ns per wave instruction per SIMD (%.2f counter ticks per instruction of one wave)\n", name, threads / 256,
           ms * 1e6 / per_simd, (double)c / (iters * 32.0));
    hipFree(out); hipFree(cyc);
}
#define R(OP, NAME) run<OP>(NAME, 256); run<OP>(NAME, 512); run<OP>(NAME, 1024);
int main() {
    R(0, "v_fma_f32") R(13, "v_mul_f32") R(1, "v_pk_fma_f32") R(12, "v_pk_mul_f32") R(2, "v_pk_fma_f16") R(7, "v_pk_mul_f16") R(9, "v_pk_min_f16")
    R(3, "v_exp_f32") R(4, "v_rcp_f32") R(10, "v_exp_f16") R(11, "v_rcp_f16") R(5, "v_cvt_pkrtz_f16_f32") R(8, "v_cvt_f32_f16") R(6, "v_med3_f32")
    return 0;
}
