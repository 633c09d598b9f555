// rb_bench — microbenchmark of the fused RawNet2 128-channel block kernel with per-phase cycle totals (developer tool)
//   tools/rb_bench [T]      (built by tools/build_rb_bench.sh with -DSVHIP_GEMM_DEBUG)
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
    const int B = 256, T = argc > 1 ? atoi(argv[1]) : 10583;
    const int gate = argc > 2 ? atoi(argv[2]) : 0;
    const int dflag = argc > 3 ? atoi(argv[3]) : 0;
    void *X, *O, *W1, *W2; float *v, *g, *part; unsigned long long* dbg;
    const int nt = rn_block128_ntiles(T), np = 4 * (nt + 1);
    CK(hipMalloc(&X, (size_t)B * T * 128 * 2)); CK(hipMalloc(&O, (size_t)B * (T / 3) * 128 * 2));
    CK(hipMalloc(&W1, 128 * 384 * 2)); CK(hipMalloc(&W2, 128 * 384 * 2)); CK(hipMalloc(&v, 4096 * 4)); CK(hipMalloc(&g, (size_t)B * 128 * 4));
    CK(hipMalloc(&part, (size_t)B * np * 128 * 4)); CK(hipMalloc(&dbg, 256 * 12 * 8));
    fill_bf16<<<2048, 256>>>((uint16_t*)X, (size_t)B * T * 128, 1, 1.0f);
    fill_bf16<<<64, 256>>>((uint16_t*)W1, 128 * 384, 2, 0.05f); fill_bf16<<<64, 256>>>((uint16_t*)W2, 128 * 384, 3, 0.05f);
    fill_f32<<<16, 256>>>(v, 4096, 0.5f); fill_f32<<<64, 256>>>(g, (size_t)B * 128, 0.7f);
    CK(hipDeviceSynchronize());
    RnBlock128Params p;
    p.xin = (const bf16_t*)X; p.bn1_scale = v; p.bn1_shift = v; p.W1 = (const bf16_t*)W1; p.W2 = (const bf16_t*)W2; p.bn2_scale = v; p.bn2_shift = v;
    if (gate) { p.alpha = v; p.gate = g; }
    p.debug = dflag;
    p.opool = (bf16_t*)O; p.colsum = part; p.B = B; p.T = T; p.Tout = T / 3; p.ntiles = nt;
    hipStream_t st; CK(hipStreamCreate(&st)); hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rd = 0; rd < 3; ++rd) {
        p.dbg = nullptr;
        for (int i = 0; i < 2; ++i) CK(launch_rn_block128(p, 256, st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 10; ++i) CK(launch_rn_block128(p, 256, st));
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
        printf("rn_block128 T=%d gate=%d  %8.3f ms  %7.1f TFLOP/s  (%d items, %.1f per CU)\n", T, gate, ms, 2.0 * 2 * B * (double)T * 128 * 384 / ms / 1e9, B * nt, B * nt / 256.0);
    }
    p.dbg = dbg;
    CK(hipMemset(dbg, 0, 256 * 12 * 8));
    CK(launch_rn_block128(p, 256, st)); CK(hipStreamSynchronize(st));
    std::vector<unsigned long long> h(256 * 12);
    CK(hipMemcpy(h.data(), dbg, 256 * 12 * 8, hipMemcpyDeviceToHost));
    const double rounds = B * nt / 256.0 + 1;
    const char* names[6] = {"phase1 work", "phase1 wait", "phase2 work", "phase2 wait", "p1 conv", "p2 conv"};
    for (int grp = 0; grp < 2; ++grp) {
        printf("group %c (wave %d) cycles per round:", grp ? 'B' : 'A', grp * 4);
        double tot = 0;
        for (int i = 0; i < 6; ++i) { double s = 0; for (int w = 0; w < 256; ++w) s += (double)h[(w * 2 + grp) * 6 + i]; s /= 256 * rounds; tot += s; printf("  %s %.0f", names[i], s); }
        printf("  | total %.0f\n", tot);
    }
    return 0;
}
