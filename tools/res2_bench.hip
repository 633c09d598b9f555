// res2_bench — microbenchmark / ablation of the fused Res2Net chain kernel (developer tool; see tools/build_gemm_bench.sh)
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
int main(int argc, char** argv) {
    const int B = 256, T = 401, C = argc > 2 ? atoi(argv[2]) : 1024, CW = C / 8;
    std::vector<int> debugs;
    { const char* d = argc > 1 ? argv[1] : "0"; while (*d) { debugs.push_back(atoi(d)); while (*d && *d != ',') ++d; if (*d == ',') ++d; } }
    void *H1, *H2, *W; float* v;
    CK(hipMalloc(&H1, (size_t)B * T * C * 2)); CK(hipMalloc(&H2, (size_t)B * T * C * 2)); CK(hipMalloc(&W, (size_t)7 * CW * 3 * CW * 2)); CK(hipMalloc(&v, 4096 * 4));
    fill_bf16<<<2048, 256>>>((uint16_t*)H1, (size_t)B * T * C, 1, 1.0f); fill_bf16<<<256, 256>>>((uint16_t*)W, (size_t)7 * CW * 3 * CW, 2, 0.05f); fill_f32<<<16, 256>>>(v, 4096, 0.5f);
    CK(hipDeviceSynchronize());
    Res2Params p; p.H1 = H1; p.H2 = H2; p.ld = C; p.T = T; p.dil = 2; p.Kp = 3 * CW;
    for (int j = 0; j < 7; ++j) { p.W[j] = (char*)W + (size_t)j * CW * 3 * CW * 2; p.bias[j] = v; p.scale[j] = v; p.shift[j] = v; }
    hipStream_t st; CK(hipStreamCreate(&st)); hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rd = 0; rd < 3; ++rd) for (int dbg : debugs) {
        p.debug = dbg;
        for (int i = 0; i < 2; ++i) CK(launch_res2net_chain(p, B, C, st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 10; ++i) CK(launch_res2net_chain(p, B, C, st));
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
        printf("res2net chain C=%d dbg %3d  %8.3f ms  %7.1f TFLOP/s\n", C, dbg, ms, 2.0 * 7 * B * T * CW * 3.0 * CW / ms / 1e9);
        if (dbg & 64) {          // stage stamps: cycles per phase of each of the seven stages, mean / max over the workgroups
            unsigned long long* dts; CK(hipMalloc(&dts, (size_t)B * 32 * 8)); CK(hipMemset(dts, 0, (size_t)B * 32 * 8));
            p.ts = dts;
            CK(launch_res2net_chain(p, B, C, st)); CK(hipStreamSynchronize(st));
            std::vector<unsigned long long> h((size_t)B * 32);
            CK(hipMemcpy(h.data(), dts, (size_t)B * 32 * 8, hipMemcpyDeviceToHost));
            p.ts = nullptr; CK(hipFree(dts));
            const char* nm[3] = {"3 taps (fragment reads + MFMAs + weight-slab barriers)", "epilogue (bias, ReLU, BN -> U) + barrier", "row pass (y -> HBM, U += c) + barrier"};
            double tot = 0;
            for (int s = 1; s < 8; ++s)
                for (int k = 0; k < 3; ++k) {
                    double sum = 0, mx = 0;
                    for (int w = 0; w < B; ++w) { const double v = (double)h[(size_t)w * 32 + s * 4 + k]; sum += v; if (v > mx) mx = v; }
                    printf("    stage %d  %-58s mean %8.0f cycles  max %8.0f\n", s, nm[k], sum / B, mx);
                    tot += sum / B;
                }
            printf("    sum of the stage means %.0f cycles = %.1f us at the clock the launch held (%.3f ms per launch: %.2f GHz)\n", tot, ms * 1e3, ms, tot / (ms * 1e6));
        }
    }
    return 0;
}
