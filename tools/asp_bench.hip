// asp_bench — per-phase timing of the fused attentive-statistics kernel (developer tool; see tools/build_gemm_bench.sh)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "kernels.h"
#include "common.h"
using namespace svhip;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
__global__ void fill_bf16(uint16_t* p, size_t n, uint32_t seed, float scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u + seed; x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
        float f = ((x & 0xffff) / 65536.0f - 0.5f) * 2.0f * scale;
        uint32_t u = __float_as_uint(f); p[i] = (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16);
    }
}
__global__ void fill_f32(float* p, size_t n, float v) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v; }
int main() {
    const int B = 256, T = 401, C = 3072;
    void *att, *W, *X; float *v, *pool; unsigned long long* dbg;
    CK(hipMalloc(&att, (size_t)B * T * 128 * 2)); CK(hipMalloc(&W, (size_t)C * 128 * 2)); CK(hipMalloc(&X, (size_t)B * T * C * 2));
    CK(hipMalloc(&v, 2 * C * 4)); CK(hipMalloc(&pool, (size_t)B * 2 * C * 4)); CK(hipMalloc(&dbg, (size_t)B * 4 * 8));
    fill_bf16<<<1024, 256>>>((uint16_t*)att, (size_t)B * T * 128, 1, 1.0f); fill_bf16<<<256, 256>>>((uint16_t*)W, (size_t)C * 128, 2, 0.1f);
    fill_bf16<<<2048, 256>>>((uint16_t*)X, (size_t)B * T * C, 3, 1.0f); fill_f32<<<16, 256>>>(v, 2 * C, 0.5f);
    CK(hipDeviceSynchronize());
    AspFusedParams p; p.att = att; p.W = W; p.bias = v; p.X = X; p.bn_scale = v; p.bn_shift = v; p.pooled_bn = pool; p.ldx = C; p.T = T; p.C = C; p.Kp = 128;
    hipStream_t st; CK(hipStreamCreate(&st)); hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rd = 0; rd < 3; ++rd) {
        for (int i = 0; i < 2; ++i) CK(launch_asp_fused(p, B, st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 10; ++i) CK(launch_asp_fused(p, B, st));
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
        printf("asp_fused  %8.3f ms\n", ms);
    }
    p.dbg = dbg;
    CK(launch_asp_fused(p, B, st)); CK(hipStreamSynchronize(st));
    std::vector<unsigned long long> h((size_t)B * 4);
    CK(hipMemcpy(h.data(), dbg, (size_t)B * 32, hipMemcpyDeviceToHost));
    double s[4] = {0, 0, 0, 0};
    for (int b = 0; b < B; ++b) for (int i = 0; i < 4; ++i) s[i] += (double)h[(size_t)b * 4 + i];
    const double tot = s[0] + s[1] + s[2] + s[3];
    printf("wave-0 ticks per workgroup (24 passes): weights+init %.0f (%.0f%%) | logit MFMAs %.0f (%.0f%%) | softmax %.0f (%.0f%%) | moments %.0f (%.0f%%) | total %.0f\n",
           s[0] / B, 100 * s[0] / tot, s[1] / B, 100 * s[1] / tot, s[2] / B, 100 * s[2] / tot, s[3] / B, 100 * s[3] / tot, tot / B);
    return 0;
}
