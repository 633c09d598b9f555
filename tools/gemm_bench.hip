// gemm_bench — microbenchmark of the library's GEMM launcher on the ECAPA layer shapes (developer tool).
//   hipcc -O3 --offload-arch=gfx950 -I speakerverification_amd/csrc tools/gemm_bench.hip speakerverification_amd/csrc/gemm.o -o tools/gemm_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <cmath>
#include "kernels.h"
#include "common.h"
#ifdef GEMM_BENCH_VENDOR
#include <hipblaslt/hipblaslt.h>
#endif
using namespace svhip;

__global__ void fill_bf16(uint16_t* p, size_t n, uint32_t seed, float scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u + seed; x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
        float f = ((x & 0xffff) / 65536.0f - 0.5f) * 2.0f * scale;
        uint32_t u = __float_as_uint(f); p[i] = (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16);
    }
}
__global__ void fill_f32(float* p, size_t n, uint32_t seed, float scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u + seed; x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
        p[i] = ((x & 0xffff) / 65536.0f - 0.5f) * 2.0f * scale;
    }
}
__global__ void checksum_bf16(const uint16_t* y, size_t n, double* out) {
    double s = 0, a = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = __uint_as_float((uint32_t)y[i] << 16);
        s += v; a += fabs((double)v) * (double)((i % 97) + 1);
    }
    atomicAdd(out, s); atomicAdd(out + 1, a);
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

#ifdef GEMM_BENCH_VENDOR
// Yardstick (round 6): the vendor's plain bf16 GEMM of the same shape on the same random operands, in the same process.
// Y[M, N] (bf16, row-major) = A[M, K] . W[N, K]^T, fp32 accumulate, no epilogue.  Column-major view: Y^T (N x M) = op_T(W: K x N) . (A^T: K x M).
// Tools only: nothing of the product links hipBLASLt.
#define LT(x) do { hipblasStatus_t s_ = (x); if (s_ != HIPBLAS_STATUS_SUCCESS) { printf("hipBLASLt error %d at %s:%d\n", (int)s_, __FILE__, __LINE__); exit(1);} } while (0)
struct VendorGemm {
    hipblasLtHandle_t lt = nullptr; void* ws = nullptr; size_t ws_bytes = 256u << 20;
    hipblasLtMatmulDesc_t desc = nullptr; hipblasLtMatrixLayout_t lw = nullptr, la = nullptr, ly = nullptr;
    hipblasLtMatmulHeuristicResult_t algo[16]; int nalgo = 0, best = 0;
    void init() { LT(hipblasLtCreate(&lt)); CK(hipMalloc(&ws, ws_bytes)); }
    void plan(int M, int N, int K) {
        if (desc) { hipblasLtMatmulDescDestroy(desc); hipblasLtMatrixLayoutDestroy(lw); hipblasLtMatrixLayoutDestroy(la); hipblasLtMatrixLayoutDestroy(ly); }
        LT(hipblasLtMatmulDescCreate(&desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
        hipblasOperation_t opT = HIPBLAS_OP_T, opN = HIPBLAS_OP_N;
        LT(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opT, sizeof(opT)));
        LT(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opN, sizeof(opN)));
        LT(hipblasLtMatrixLayoutCreate(&lw, HIP_R_16BF, K, N, K));
        LT(hipblasLtMatrixLayoutCreate(&la, HIP_R_16BF, K, M, K));
        LT(hipblasLtMatrixLayoutCreate(&ly, HIP_R_16BF, N, M, N));
        hipblasLtMatmulPreference_t pref; LT(hipblasLtMatmulPreferenceCreate(&pref));
        LT(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws_bytes, sizeof(ws_bytes)));
        LT(hipblasLtMatmulAlgoGetHeuristic(lt, desc, lw, la, ly, ly, pref, 16, algo, &nalgo));
        hipblasLtMatmulPreferenceDestroy(pref);
        if (nalgo <= 0) { printf("hipBLASLt: no algorithm for %d x %d x %d\n", M, N, K); exit(1); }
        best = 0;
    }
    hipblasStatus_t run(const void* A, const void* W, void* Y, hipStream_t st, int which) {
        const float one = 1.0f, zero = 0.0f;
        return hipblasLtMatmul(lt, desc, &one, W, lw, A, la, &zero, Y, ly, Y, ly, &algo[which].algo, ws, ws_bytes, st);
    }
    // time every heuristic candidate once (3 launches each) and keep the fastest: the yardstick is the vendor's best, not its first guess
    void tune(const void* A, const void* W, void* Y, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
        float bt = 1e30f;
        for (int a = 0; a < nalgo; ++a) {
            if (run(A, W, Y, st, a) != HIPBLAS_STATUS_SUCCESS) continue;
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < 3; ++i) run(A, W, Y, st, a);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < bt) { bt = ms; best = a; }
        }
    }
};
#endif

struct Shape { const char* name; int M, N, K, taps, dil, cin; int act1, act2; bool a2; bool out_f32; int T = 0; int pad = PAD_REFLECT; bool resid = false; };

int main(int argc, char** argv) {
    const bool bf16 = !(argc > 1 && atoi(argv[1]) == 0);
    const int B = argc > 2 ? atoi(argv[2]) : 256;
    std::vector<int> debugs;                       // comma-separated list: the variants run interleaved in this process (A/B)
    { const char* d = argc > 3 ? argv[3] : "0"; while (*d) { debugs.push_back(atoi(d)); while (*d && *d != ',') ++d; if (*d == ',') ++d; } }
    const int rounds = argc > 5 ? atoi(argv[5]) : 1;
    const float dscale = argc > 4 ? (float)atof(argv[4]) : 1.0f;
    const int T = 401, M = B * T, C = 1024;
    const int esz = bf16 ? 2 : 4;
    std::vector<Shape> shapes = {
        {"tdnn  N1024 K1024 gelu", M, C, C, 1, 1, 0, ACT_GELU, ACT_NONE, false, false},
        {"mfa   N3072 K3072 gelu", M, 3 * C, 3 * C, 1, 1, 0, ACT_GELU, ACT_NONE, false, false},
        {"asp_t N128  K3072 relu-tanh", M, 128, 3 * C, 1, 1, 0, ACT_RELU, ACT_TANH, false, false},
        {"asp_c N3072 K128  f32out", M, 3 * C, 128, 1, 1, 0, ACT_NONE, ACT_NONE, false, true},
        {"blk0  N1024 K400 conv5 gelu", M, C, 400, 5, 1, 80, ACT_GELU, ACT_NONE, false, false},
        {"res2  N128  K384 conv3d2 relu +A2", M, 128, 384, 3, 2, 128, ACT_RELU, ACT_NONE, true, false},
        {"tdnn  N1024 K1024 none", M, C, C, 1, 1, 0, ACT_NONE, ACT_NONE, false, false},
    };
    if (argc > 6 && !strcmp(argv[6], "rn")) {          // RawNet2 blocks 2 - 7 at B utterances of 32 000 samples (k = 3 convs, zero padding)
        const int T2 = 1175, T3 = 391, T5 = 130, T6 = 43;
        shapes = {
            {"rn b2 conv1 N256 K384 c3 bn-lrelu", B * T2, 256, 384, 3, 1, 128, ACT_NONE, ACT_LRELU03, false, false, T2, PAD_ZERO},
            {"rn b2 conv2 N256 K768 c3", B * T2, 256, 768, 3, 1, 256, ACT_NONE, ACT_NONE, false, false, T2, PAD_ZERO},
            {"rn b2 plain N256 K768", B * T2, 256, 768, 1, 1, 0, ACT_NONE, ACT_NONE, false, false, T2, PAD_ZERO},
            {"rn b2 conv2 +R N256 K768 c3", B * T2, 256, 768, 3, 1, 256, ACT_NONE, ACT_NONE, false, false, T2, PAD_ZERO, true},
            {"rn b2 plain +R N256 K768", B * T2, 256, 768, 1, 1, 0, ACT_NONE, ACT_NONE, false, false, T2, PAD_ZERO, true},
            {"rn b3 conv  N256 K768 c3 bn-lrelu", B * T3, 256, 768, 3, 1, 256, ACT_NONE, ACT_LRELU03, false, false, T3, PAD_ZERO},
            {"rn b3 plain N256 K768", B * T3, 256, 768, 1, 1, 0, ACT_NONE, ACT_NONE, false, false, T3, PAD_ZERO},
            {"rn b5 conv2 N512 K1536 c3", B * T5, 512, 1536, 3, 1, 512, ACT_NONE, ACT_NONE, false, false, T5, PAD_ZERO},
            {"rn b6 conv  N512 K1536 c3 bn-lrelu", B * T6, 512, 1536, 3, 1, 512, ACT_NONE, ACT_LRELU03, false, false, T6, PAD_ZERO},
            {"rn b5 conv2 +R N512 K1536 c3", B * T5, 512, 1536, 3, 1, 512, ACT_NONE, ACT_NONE, false, false, T5, PAD_ZERO, true},
            {"rn b6 conv2 +R N512 K1536 c3", B * T6, 512, 1536, 3, 1, 512, ACT_NONE, ACT_NONE, false, false, T6, PAD_ZERO, true},
            {"rn b6 plain +R N512 K1536", B * T6, 512, 1536, 1, 1, 0, ACT_NONE, ACT_NONE, false, false, T6, PAD_ZERO, true},
            {"rn b6 plain N512 K1536", B * T6, 512, 1536, 1, 1, 0, ACT_NONE, ACT_NONE, false, false, T6, PAD_ZERO},
        };
    }
    size_t maxA = (size_t)M * 3 * C, maxW = (size_t)3 * C * 3 * C + 128 * 3 * C, maxY = (size_t)M * 3 * C;
    void *A, *A2, *W, *Y; float *bias, *scale, *shift;
    CK(hipMalloc(&A, maxA * esz + 256)); CK(hipMemset((char*)A + maxA * esz, 0, 256)); CK(hipMalloc(&A2, maxA * esz)); CK(hipMalloc(&W, maxW * esz)); CK(hipMalloc(&Y, maxY * 4));
    CK(hipMalloc(&bias, 4096 * 4)); CK(hipMalloc(&scale, 4096 * 4)); CK(hipMalloc(&shift, 4096 * 4));
    if (bf16) { fill_bf16<<<2048, 256>>>((uint16_t*)A, maxA, 1, dscale); fill_bf16<<<2048, 256>>>((uint16_t*)A2, maxA, 2, dscale); fill_bf16<<<2048, 256>>>((uint16_t*)W, maxW, 3, 0.05f * dscale); }
    else { fill_f32<<<2048, 256>>>((float*)A, maxA, 1, 1.0f); fill_f32<<<2048, 256>>>((float*)A2, maxA, 2, 1.0f); fill_f32<<<2048, 256>>>((float*)W, maxW, 3, 0.05f); }
    fill_f32<<<16, 256>>>(bias, 4096, 4, 0.1f); fill_f32<<<16, 256>>>(scale, 4096, 5, 1.0f); fill_f32<<<16, 256>>>(shift, 4096, 6, 0.1f);
    void* zp = (char*)A + maxA * esz;      // zero page BEHIND the A operand (gemm_pw3's 16-bit conv-gather form addresses it as a 32-bit offset)
    float* csum; const int64_t csr = (int64_t)(M / 256 + 2) * 16 * 3 * C; CK(hipMalloc(&csum, 2 * csr * 4));
    CK(hipDeviceSynchronize());
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
#ifdef GEMM_BENCH_VENDOR
    VendorGemm vg; vg.init();
#endif
    for (auto& s : shapes) {
        GemmParams p;
        p.A = A; p.A2 = s.a2 ? A2 : nullptr; p.W = W; p.Y = Y; p.bias = bias; p.scale = scale; p.shift = shift;
        p.M = s.M; p.N = s.N; p.K = s.K; p.Kp = round_up(s.K, gemm_bk(bf16)); p.Wrows = round_up(s.N, 128);
        p.lda = s.taps > 1 ? s.cin * (s.a2 ? 8 : 1) : s.K; p.lda2 = p.lda; p.ldy = s.N; p.T = s.T ? s.T : T;
        p.taps = s.taps; p.dil = s.dil; p.cin = s.cin; p.pad_mode = s.pad; p.act1 = s.act1; p.act2 = s.act2; p.out_f32 = s.out_f32; p.debug = 0; p.zero_page = zp;
        if (s.resid) { p.R = A2; p.ldr = s.N; }
#ifdef GEMM_BENCH_VENDOR
        bool vplanned = false;
#endif
        for (int rd = 0; rd < rounds; ++rd)
        for (int debug : debugs) {
#ifdef GEMM_BENCH_VENDOR
            if (debug == (1 << 20)) {                    // 1048576: the vendor GEMM of this shape (pointwise bf16 shapes only)
                if (!bf16 || s.taps != 1 || s.out_f32) continue;
                if (!vplanned) { vg.plan(s.M, s.N, s.K); vg.tune(A, W, Y, st, e0, e1); vplanned = true; }
                for (int i = 0; i < 2; ++i) vg.run(A, W, Y, st, vg.best);
                CK(hipStreamSynchronize(st));
                const int it = 10;
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < it; ++i) vg.run(A, W, Y, st, vg.best);
                CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= it;
                printf("%-36s vendor     %8.3f ms  %8.1f TFLOP/s  [hipBLASLt algo %d of %d, no epilogue]\n", s.name, ms, 2.0 * s.M * s.N * s.K / ms / 1e9, vg.best, vg.nalgo);
                continue;
            }
#endif
            p.debug = debug;
            p.cv_off = (debug & 32768) ? 1 : 0;            // 32768: conv-gather shapes on the per-tile kernel
            p.tail_split = (debug & 65536) ? 0 : 1;        // 65536: the persistent kernel's last partial round as whole tiles (round 4)
            if (debug & 128) { p.colsum = csum; p.colsum_sq = (debug & 256) ? 1 : 0; p.colsum_stride = csr; } else { p.colsum = nullptr; }
            const bool pw4 = (debug & (1 << 21)) && bf16 && gemm_pw4_supported(p, bf16) && gemm_route(p, bf16) == ROUTE_PW3;      // 2097152: the four-wave kernel
            auto launch = [&]() { return pw4 ? launch_gemm_pw4(p, st) : launch_gemm(p, bf16, st); };
            if ((debug & (1 << 21)) && !pw4) continue;
            for (int i = 0; i < 2; ++i) CK(launch());
            CK(hipStreamSynchronize(st));
            const int it = 10;
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < it; ++i) CK(launch());
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= it;
            double hcs[2] = {0, 0};
            if (bf16 && !s.out_f32) {          // order-sensitive checksum of the output: variants of one shape must agree
                static double* dcs = nullptr;
                if (!dcs) CK(hipMalloc(&dcs, 16));
                CK(hipMemsetAsync(dcs, 0, 16, st));
                checksum_bf16<<<1024, 256, 0, st>>>((const uint16_t*)Y, (size_t)s.M * s.N, dcs);
                CK(hipMemcpyAsync(hcs, dcs, 16, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
            }
            printf("%-36s dbg %5d %8.3f ms  %8.1f TFLOP/s  cs %.6e %.6e  [%s]\n", s.name, debug, ms, 2.0 * s.M * s.N * s.K / ms / 1e9, hcs[0], hcs[1],
                   pw4 ? "pw4" : gemm_route(p, bf16) == ROUTE_PW3 ? "pw3" : gemm_route(p, bf16) == ROUTE_PW3CV ? "pw3cv16" : gemm_route(p, bf16) == ROUTE_PW2 ? "pw2" : "other");
            if ((debug & 128) && p.colsum && gemm_pw2_supported(p, bf16)) {
                // per-utterance column sums from the partials (the arithmetic of colsum_finalize_kernel), as an order-sensitive checksum:
                // the 8-row-group layout of pw2 and the 2-row-group layout of pw3 must agree to bf16 rounding of the summed values
                const int RG = gemm_colsum_groups(p, bf16), Tn = p.T, Bn = s.M / Tn;
                const size_t nfl = (size_t)2 * csr;
                std::vector<float> hp(nfl);
                CK(hipMemcpy(hp.data(), csum, nfl * 4, hipMemcpyDeviceToHost));
                double a1 = 0, a2 = 0, q1 = 0;
                for (int b = 0; b < Bn; ++b) {
                    const int r0 = b * Tn, r1 = r0 + Tn - 1;
                    for (int n = 0; n < s.N; n += 7) {
                        double sm = 0, sq = 0;
                        for (int tm = r0 / 256; tm <= r1 / 256; ++tm) {
                            const int seg = b - (tm * 256) / Tn;
                            if (seg < 0 || seg > 1) continue;
                            for (int rg = 0; rg < RG; ++rg) {
                                const size_t o = ((size_t)(tm * RG + rg) * 2 + seg) * s.N + n;
                                sm += hp[o];
                                if (debug & 256) sq += hp[csr + o];
                            }
                        }
                        a1 += sm * ((b * 31 + n) % 13 + 1); a2 += fabs(sm); q1 += sq * ((b + n) % 5 + 1);
                    }
                }
                printf("    colsum (RG %d): weighted sum %.9e  abs sum %.9e  sq %.9e\n", RG, a1, a2, q1);
            }
            if ((debug & 16384) && (gemm_route(p, bf16) == ROUTE_PW3 || gemm_route(p, bf16) == ROUTE_PW3CV)) {      // stage cycle totals of the persistent kernel, per wave
                const int nwg = 256;
                unsigned long long* dts; CK(hipMalloc(&dts, (size_t)nwg * 512)); CK(hipMemset(dts, 0, (size_t)nwg * 512));
                p.ts = dts;
                CK(launch_gemm(p, bf16, st)); CK(hipStreamSynchronize(st));
                std::vector<unsigned long long> hts((size_t)nwg * 64);
                CK(hipMemcpy(hts.data(), dts, (size_t)nwg * 512, hipMemcpyDeviceToHost));
                p.ts = nullptr; CK(hipFree(dts));
                for (int wv = 0; wv < 8; ++wv) {
                    double sum[4] = {0, 0, 0, 0}, tot = 0, tiles = 0; int nw = 0;
                    for (int w = 0; w < nwg; ++w) {
                        const unsigned long long* o = &hts[((size_t)w * 8 + wv) * 8];
                        if (!o[4]) continue;
                        ++nw; tiles += (double)o[4]; tot += (double)o[5];
                        for (int i = 0; i < 4; ++i) sum[i] += (double)o[i];
                    }
                    if (tiles > 0)
                        printf("    pw3 wave %d: %d WGs, %.2f tiles/WG; cycles per tile: start wait %.0f | K loop %.0f | next-tile issue %.0f | epilogue %.0f | sum %.0f (kernel %.0f per WG = %.2f GHz)\n",
                               wv, nw, tiles / nw, sum[0] / tiles, sum[1] / tiles, sum[2] / tiles, sum[3] / tiles, (sum[0] + sum[1] + sum[2] + sum[3]) / tiles, tot / nw,
                               tot / nw / (ms * 1e6));
                }
                continue;
            }
            if (debug & 16384) {          // stage timestamps of one launch (non-persistent pw2 kernel)
                const int nwg = ((s.M + 255) / 256) * ((s.N + 255) / 256);
                unsigned long long* dts; CK(hipMalloc(&dts, (size_t)nwg * 64)); CK(hipMemset(dts, 0, (size_t)nwg * 64));
                p.ts = dts;
                CK(launch_gemm(p, bf16, st)); CK(hipStreamSynchronize(st));
                std::vector<unsigned long long> hts((size_t)nwg * 8);
                CK(hipMemcpy(hts.data(), dts, (size_t)nwg * 64, hipMemcpyDeviceToHost));
                p.ts = nullptr; CK(hipFree(dts));
                unsigned long long t0 = ~0ull, t1 = 0;      // s_memtime is per XCD: calibrate on the XCD of workgroup 0
                const unsigned xcc0 = (unsigned)hts[7] & 15;
                double sum[5] = {0, 0, 0, 0, 0};
                for (int w = 0; w < nwg; ++w) {
                    const unsigned long long* o = &hts[(size_t)w * 8];
                    if (((unsigned)o[7] & 15) == xcc0) { if (o[0] < t0) t0 = o[0]; if (o[5] > t1) t1 = o[5]; }
                    for (int i = 0; i < 5; ++i) sum[i] += (double)(o[i + 1] - o[i]);
                }
                const double span = (double)(t1 - t0), tick_us = ms * 1e3 / span;      // calibrate ticks with the event time
                printf("    stamps: span %.0f ticks (%.5f us/tick); mean per WG [us]: prologue %.2f | main %.2f | epi-compute+stage %.2f | colsum %.2f | stores+drain %.2f | total %.2f\n",
                       span, tick_us, sum[0] / nwg * tick_us, sum[1] / nwg * tick_us, sum[2] / nwg * tick_us, sum[3] / nwg * tick_us, sum[4] / nwg * tick_us,
                       (sum[0] + sum[1] + sum[2] + sum[3] + sum[4]) / nwg * tick_us);
                // gaps between consecutive workgroups on the same CU slot: sort by (xcc, hw_id cu/se bits)
                std::vector<std::pair<unsigned long long, int>> order;
                std::vector<std::vector<int>> percu(8 * 4096);
                for (int w = 0; w < nwg; ++w) {
                    const unsigned long long* o = &hts[(size_t)w * 8];
                    const unsigned hw = (unsigned)o[6], xcc = (unsigned)o[7] & 15;
                    const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;      // gfx9 HW_ID: CU_ID[11:8] SH_ID[12] SE_ID[15:13]
                    percu[(xcc * 8 + se) * 32 + sh * 16 + cu].push_back(w);
                }
                double gap = 0; int ngap = 0, ncu = 0; double busy = 0;
                for (auto& v : percu) {
                    if (v.empty()) continue;
                    ++ncu;
                    std::sort(v.begin(), v.end(), [&](int a, int b) { return hts[(size_t)a * 8] < hts[(size_t)b * 8]; });
                    for (size_t i = 0; i < v.size(); ++i) {
                        busy += (double)(hts[(size_t)v[i] * 8 + 5] - hts[(size_t)v[i] * 8]);
                        if (i) { gap += (double)hts[(size_t)v[i] * 8] - (double)hts[(size_t)v[i - 1] * 8 + 5]; ++ngap; }
                    }
                }
                printf("    %d CU slots seen, mean gap between successive WGs on a CU %.2f us, mean busy per CU %.1f us of %.1f us\n",
                       ncu, ngap ? gap / ngap * tick_us : 0.0, busy / ncu * tick_us, span * tick_us);
            }
        }
    }
    return 0;
}
