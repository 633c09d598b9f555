// r2_bench — the Res2Net step form of gemm_pw3 (F32X3 handles) alone, with the kernel's stage stamps (developer tool).
//   bash tools/build_gemm_bench.sh builds it next to gemm_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "kernels.h"
#include "common.h"
using namespace svhip;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void fill_u16(uint16_t* p, size_t n, uint32_t seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u + seed; x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13;
        float f = ((x & 0xffff) / 65536.0f - 0.5f) * 0.2f;
        uint32_t u = __float_as_uint(f); p[i] = (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16);
    }
}
__global__ void fill_f(float* p, size_t n, uint32_t seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u + seed; x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13;
        p[i] = ((x & 0xffff) / 65536.0f - 0.5f);
    }
}
static int stamps(GemmParams p, hipStream_t st, bool x3) {
    const int nwg = 256;
    unsigned long long* dts; CK(hipMalloc(&dts, (size_t)nwg * 512)); CK(hipMemset(dts, 0, (size_t)nwg * 512));
    p.ts = dts; p.debug = 16384;
    CK(x3 ? launch_gemm_pw3x3(p, st) : launch_gemm_pw3r2(p, st)); CK(hipStreamSynchronize(st));
    std::vector<unsigned long long> h((size_t)nwg * 64);
    CK(hipMemcpy(h.data(), dts, (size_t)nwg * 512, hipMemcpyDeviceToHost));
    for (int wv = 0; wv < 8; ++wv) {
        double sum[4] = {0, 0, 0, 0}, tot = 0, tiles = 0; int nw = 0;
        for (int w = 0; w < nwg; ++w) {
            const unsigned long long* o = &h[((size_t)w * 8 + wv) * 8];
            if (!o[4]) continue;
            ++nw; tiles += (double)o[4]; tot += (double)o[5];
            for (int i = 0; i < 4; ++i) sum[i] += (double)o[i];
        }
        if (tiles > 0) printf("  wave %d: %d WGs, %.2f tiles/WG; cycles per tile: pre-loop %.0f | K loop %.0f | next-tile issue %.0f | epilogue %.0f; kernel %.0f cycles per WG\n",
                              wv, nw, tiles / nw, sum[0] / tiles, sum[1] / tiles, sum[2] / tiles, sum[3] / tiles, tot / nw);
    }
    CK(hipFree(dts));
    return 0;
}

// the pointwise X3 form (tdnn1 / tdnn2 / mfa of an F32X3 handle) on random S32 operands
static int run_x3(int B, int N, int K, int cs) {
    const int T = 401, M = B * T;
    void *A, *W; float *Y, *bias, *scale, *shift, *colsum = nullptr;
    CK(hipMalloc(&A, (size_t)M * K * 4)); CK(hipMalloc(&W, (size_t)N * K * 4)); CK(hipMalloc(&Y, (size_t)M * N * 4));
    CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&scale, N * 4)); CK(hipMalloc(&shift, N * 4));
    fill_u16<<<2048, 256>>>((uint16_t*)A, (size_t)M * K * 2, 1); fill_u16<<<2048, 256>>>((uint16_t*)W, (size_t)N * K * 2, 2);
    fill_f<<<4, 256>>>(bias, N, 4); fill_f<<<4, 256>>>(scale, N, 5); fill_f<<<4, 256>>>(shift, N, 6);
    const int64_t region = (int64_t)((M + 255) / 256 + 2) * 16 * N;
    if (cs) { CK(hipMalloc(&colsum, (size_t)4 * region * 4)); CK(hipMemset(colsum, 0, (size_t)4 * region * 4)); }
    CK(hipDeviceSynchronize());
    hipStream_t st; CK(hipStreamCreate(&st));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    GemmParams p;
    p.A = A; p.lda = K; p.W = W; p.Wrows = N; p.x3 = 2; p.bias = bias; p.scale = scale; p.shift = shift;
    p.M = M; p.N = N; p.K = K; p.Kp = K; p.T = T; p.taps = 1; p.act1 = ACT_GELU; p.Y = Y; p.ldy = N; p.num_cu = prop.multiProcessorCount;
    if (cs) { p.colsum = colsum; p.colsum_sq = cs > 1; p.colsum_stride = region; }
    if (!gemm_pw3x3_supported(p)) { printf("x3 unsupported\n"); return 1; }
    for (int i = 0; i < 3; ++i) CK(launch_gemm_pw3x3(p, st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 10; ++i) CK(launch_gemm_pw3x3(p, st));
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    printf("x3 B=%d N=%d K=%d cs=%d: %.3f ms  (%.1f TFLOP/s of reference FLOPs = %.3f of the x3 ceiling)\n", B, N, K, cs, ms, 2.0 * M * N * K / ms / 1e9,
           2.0 * M * N * K / ms / 1e9 / 833.3);
    return stamps(p, st, true);
}

int main(int argc, char** argv) {
    if (argc > 2 && atoi(argv[2]) > 0) return run_x3(atoi(argv[1]), atoi(argv[2]), argc > 3 ? atoi(argv[3]) : atoi(argv[2]), argc > 4 ? atoi(argv[4]) : 0);
    const int B = argc > 1 ? atoi(argv[1]) : 256, T = 401, C = 1024, CW = 128, M = B * T;
    void *U, *W, *Y, *Y2; float *H1, *bias, *scale, *shift;
    CK(hipMalloc(&U, (size_t)M * CW * 4)); CK(hipMalloc(&Y2, (size_t)M * CW * 4)); CK(hipMalloc(&W, (size_t)CW * 3 * CW * 4));
    CK(hipMalloc(&Y, (size_t)M * C * 4)); CK(hipMalloc(&H1, (size_t)M * C * 4));
    CK(hipMalloc(&bias, 4096)); CK(hipMalloc(&scale, 4096)); CK(hipMalloc(&shift, 4096));
    fill_u16<<<2048, 256>>>((uint16_t*)U, (size_t)M * CW * 2, 1); fill_u16<<<256, 256>>>((uint16_t*)W, (size_t)CW * 3 * CW * 2, 2);
    fill_f<<<2048, 256>>>(H1, (size_t)M * C, 3); fill_f<<<4, 256>>>(bias, 1024, 4); fill_f<<<4, 256>>>(scale, 1024, 5); fill_f<<<4, 256>>>(shift, 1024, 6);
    CK(hipDeviceSynchronize());
    hipStream_t st; CK(hipStreamCreate(&st));
    GemmParams p;
    p.A = U; p.lda = CW; p.W = W; p.Wrows = CW; p.x3 = 2; p.bias = bias; p.scale = scale; p.shift = shift;
    p.M = M; p.N = CW; p.K = 3 * CW; p.Kp = 3 * CW; p.T = T; p.taps = 3; p.dil = 2; p.cin = CW; p.pad_mode = PAD_REFLECT;
    p.act1 = ACT_RELU; p.Y = (char*)Y + 2 * CW * 4; p.ldy = C; p.R = H1 + 3 * CW; p.ldr = C; p.Y2 = Y2; p.lda2 = CW;
    if (!gemm_pw3r2_supported(p)) { printf("unsupported\n"); return 1; }
    for (int i = 0; i < 3; ++i) CK(launch_gemm_pw3r2(p, st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 20; ++i) CK(launch_gemm_pw3r2(p, st));
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
    printf("r2 step B=%d: %.1f us  (%.1f TFLOP/s of reference FLOPs)\n", B, ms * 1e3, 2.0 * M * CW * 3 * CW / ms / 1e9);
    return stamps(p, st, false);
}
