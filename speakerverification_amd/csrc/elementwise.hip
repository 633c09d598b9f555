// elementwise.hip — frame-major reductions and elementwise kernels of the ECAPA / RawNet2 stacks.
// All of these are HBM-bound: 16-byte vector accesses, lanes along the contiguous channel axis,
// wavefront-parallel over T with a small LDS combine, fp32 statistics.
#include "common.h"
#include "kernels.h"

namespace svhip {

namespace {

// ---- mean / std over the T rows of each utterance ---------------------------------------------
// grid (ceil(C / (64*VEC)), B), block 256: wave w takes frames t = w, w+4, ...; lane owns VEC channels.
template <typename T, bool STD>
__global__ __launch_bounds__(256) void colstats_kernel(const T* __restrict__ X, int ldx, int Tn, int C,
                                                       float* __restrict__ out, int ld_out, float eps) {
    constexpr int VEC = Vec16<T>::N;
    __shared__ float red[4][64 * VEC];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = (blockIdx.x * 64 + lane) * VEC;
    const bool ok = c0 < C;
    const T* __restrict__ base = X + (int64_t)b * Tn * ldx + c0;
    float s[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) s[j] = 0.0f;
    if (ok)
        for (int t = wave; t < Tn; t += 4) {
            Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(base + (int64_t)t * ldx);
#pragma unroll
            for (int j = 0; j < VEC; ++j) s[j] += v.get(j);
        }
#pragma unroll
    for (int j = 0; j < VEC; ++j) red[wave][lane * VEC + j] = s[j];
    __syncthreads();
    float mean[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const int e = lane * VEC + j;
        mean[j] = (red[0][e] + red[1][e] + red[2][e] + red[3][e]) / (float)Tn;
    }
    if (!STD) {
        if (wave == 0 && ok)
#pragma unroll
            for (int j = 0; j < VEC; ++j) out[(int64_t)b * ld_out + c0 + j] = mean[j];
        return;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < VEC; ++j) s[j] = 0.0f;
    if (ok)
        for (int t = wave; t < Tn; t += 4) {
            Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(base + (int64_t)t * ldx);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float d = v.get(j) - mean[j];
                s[j] = fmaf(d, d, s[j]);
            }
        }
#pragma unroll
    for (int j = 0; j < VEC; ++j) red[wave][lane * VEC + j] = s[j];
    __syncthreads();
    if (wave == 0 && ok)
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const int e = lane * VEC + j;
            const float var = (red[0][e] + red[1][e] + red[2][e] + red[3][e]) / (float)Tn;
            out[(int64_t)b * ld_out + c0 + j] = mean[j];
            out[(int64_t)b * ld_out + C + c0 + j] = sqrtf(fmaxf(var, eps));
        }
}

// ---- long-T column mean in two deterministic stages (RawNet2 AFMS: T up to 10583, C = 128) -------------------
// stage 1: grid (ceil(C / (64*VEC)), B, TS): partial sums over a T-slice -> part (B, TS, C);  stage 2: sum / T.
template <typename T>
__global__ __launch_bounds__(256) void colsum_slice_kernel(const T* __restrict__ X, int ldx, int Tn, int C, int TS,
                                                           float* __restrict__ part) {
    constexpr int VEC = Vec16<T>::N;
    __shared__ float red[4][64 * VEC];
    const int b = blockIdx.y, ts = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = (blockIdx.x * 64 + lane) * VEC;
    const bool ok = c0 < C;
    const int per = (Tn + TS - 1) / TS;
    const int t0 = ts * per, t1 = min(Tn, t0 + per);
    const T* __restrict__ base = X + (int64_t)b * Tn * ldx + c0;
    float s[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) s[j] = 0.0f;
    if (ok)
        for (int t = t0 + wave; t < t1; t += 4) {
            Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(base + (int64_t)t * ldx);
#pragma unroll
            for (int j = 0; j < VEC; ++j) s[j] += v.get(j);
        }
#pragma unroll
    for (int j = 0; j < VEC; ++j) red[wave][lane * VEC + j] = s[j];
    __syncthreads();
    if (wave == 0 && ok)
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const int e = lane * VEC + j;
            part[((int64_t)b * TS + ts) * C + c0 + j] = red[0][e] + red[1][e] + red[2][e] + red[3][e];
        }
}

__global__ __launch_bounds__(256) void colsum_slice_finalize_kernel(const float* __restrict__ part, int TS, int C, int Tn,
                                                                    float* __restrict__ mean, int n) {
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= n) return;
    const int b = id / C, c = id - b * C;
    float s = 0.0f;
    for (int ts = 0; ts < TS; ++ts) s += part[((int64_t)b * TS + ts) * C + c];
    mean[id] = s / (float)Tn;
}

// ---- small-M linear: out[b, n] = act(bias[n] + W[n, :] . in[b, :]) -----------------------------
// grid (ceil(N / 8), ceil(B / 8)), block 256: an 8 (n) x 8 (b) output block per workgroup, K split
// over the 4 waves, float4 loads along K (both operands are read once per block instead of once per
// output), 64 accumulators per lane, wavefront shuffle + LDS combine at the end.
__global__ __launch_bounds__(256) void rowvec_linear_kernel(const float* __restrict__ in, int ld_in,
                                                            const float* __restrict__ W, const float* __restrict__ bias,
                                                            float* __restrict__ out, int ld_out, int B, int N, int K, int act) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = blockIdx.x * 8, b0 = blockIdx.y * 8;
    const float* __restrict__ wrow[8];
    const float* __restrict__ xrow[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        wrow[i] = W + (int64_t)min(n0 + i, N - 1) * K;
        xrow[i] = in + (int64_t)min(b0 + i, B - 1) * ld_in;
    }
    float acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = 0.0f;
    const int K4 = K >> 2;                                   // float4 chunks along K
#pragma unroll 2
    for (int c = wave * 64 + lane; c < K4; c += 256) {        // two trips' 32 loads in flight
        f32x4 wv[8], xv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            wv[i] = *reinterpret_cast<const f32x4*>(wrow[i] + 4 * c);
            xv[i] = *reinterpret_cast<const f32x4*>(xrow[i] + 4 * c);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j] = fmaf(wv[i][e], xv[j][e], acc[i][j]);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float s = wave_sum(acc[i][j]);
            if (lane == 0) red[wave][i * 8 + j] = s;
        }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int i = threadIdx.x >> 3, j = threadIdx.x & 7;     // n = n0 + i, b = b0 + j
        const int n = n0 + i, b = b0 + j;
        if (n < N && b < B) {
            const float s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
            out[(int64_t)b * ld_out + n] = apply_act(s + (bias ? bias[n] : 0.0f), act);
        }
    }
}

// Small batches (B <= 64: the reference API's per-file calls embed 10 - 20 crops): the grid above is 48 - 72 workgroups of four waves, each
// walking K in six dependent trips — 27 us for fc at B = 20 (4.7 MB of weights).  Same arithmetic with 16 waves per workgroup on an
// 8 (n) x 4 (b) block: K is covered in two trips and every CU holds more loads in flight.  The per-output sum is formed in a different
// order than in the kernel above (fp32 round-off; a batch takes ONE of the two kernels, chosen by its size alone).
__global__ __launch_bounds__(1024) void rowvec_linear_small_kernel(const float* __restrict__ in, int ld_in,
                                                                   const float* __restrict__ W, const float* __restrict__ bias,
                                                                   float* __restrict__ out, int ld_out, int B, int N, int K, int act) {
    __shared__ float red[16][32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = blockIdx.x * 8, b0 = blockIdx.y * 4;
    const float* __restrict__ wrow[8];
    const float* __restrict__ xrow[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) wrow[i] = W + (int64_t)min(n0 + i, N - 1) * K;
#pragma unroll
    for (int j = 0; j < 4; ++j) xrow[j] = in + (int64_t)min(b0 + j, B - 1) * ld_in;
    float acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0f;
    const int K4 = K >> 2;
#pragma unroll 2
    for (int c = threadIdx.x; c < K4; c += 1024) {
        f32x4 wv[8], xv[4];
#pragma unroll
        for (int i = 0; i < 8; ++i) wv[i] = *reinterpret_cast<const f32x4*>(wrow[i] + 4 * c);
#pragma unroll
        for (int j = 0; j < 4; ++j) xv[j] = *reinterpret_cast<const f32x4*>(xrow[j] + 4 * c);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j] = fmaf(wv[i][e], xv[j][e], acc[i][j]);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float s = wave_sum(acc[i][j]);
            if (lane == 0) red[wave][i * 4 + j] = s;
        }
    __syncthreads();
    if (threadIdx.x < 32) {
        const int i = threadIdx.x >> 2, j = threadIdx.x & 3;
        const int n = n0 + i, b = b0 + j;
        if (n < N && b < B) {
            float s = 0.0f;
#pragma unroll
            for (int w = 0; w < 16; ++w) s += red[w][threadIdx.x];
            out[(int64_t)b * ld_out + n] = apply_act(s + (bias ? bias[n] : 0.0f), act);
        }
    }
}

// Full batches (B > 64, K = 6 144 for ECAPA's fc / asp_ctx): the 8 x 8-output kernel above moves 300 MB from L2 for 11 MB of operands (each
// workgroup reads 8 rows of both matrices end to end): 36 - 39 us per layer.  Here the product runs on the exact fp32 MFMA
// (v_mfma_f32_32x32x2: a k-ordered fmaf chain, one rounding per product) with K split over workgroups: a workgroup = a 32 (n) x 32 (b) output
// tile over a K slice of LIN_KS = 384 — its four waves take 96 k each, a lane's 16-byte load supplies four MFMA steps (the k order inside a
// dot product is permuted the same way for both operands) — and writes its partial tile to a scratch; a second launch adds the slices in a
// fixed order, the bias and the activation.  Deterministic; a row's sums do not depend on the batch it rides in.
// LIN_KS: the K slice of a workgroup — 384 for the long rows (ECAPA's fc / asp_ctx, K = 6 144: 16 slices), 256 for rows that are a multiple
// of 256 but not of 384 (RawNet2's fc, K = 1 024: 4 slices; round 5)
template <int LIN_KS>
__global__ __launch_bounds__(256) void rowvec_linear_mfma_part_kernel(const float* __restrict__ in, int ld_in, const float* __restrict__ W,
                                                                      float* __restrict__ part, int B, int N, int K) {
    __shared__ float red[3][16][64];                   // partial tiles of waves 1 - 3: [wave][register][lane]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.x * 32, b0 = blockIdx.y * 32, ks = blockIdx.z;
    const int k0 = ks * LIN_KS + wave * (LIN_KS / 4) + 4 * h;
    const float* wrow = W + (int64_t)min(n0 + r, N - 1) * K + k0;                 // A operand: row = output channel
    const float* xrow = in + (int64_t)min(b0 + r, B - 1) * ld_in + k0;            // B operand: column = utterance
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    f32x4 wv[LIN_KS / 32], xv[LIN_KS / 32];
#pragma unroll
    for (int j = 0; j < LIN_KS / 32; ++j) {                                       // 12 groups of 8 k: every load of the slice in flight at once
        wv[j] = *reinterpret_cast<const f32x4*>(wrow + 8 * j);
        xv[j] = *reinterpret_cast<const f32x4*>(xrow + 8 * j);
    }
#pragma unroll
    for (int j = 0; j < LIN_KS / 32; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[j][e], xv[j][e], acc, 0, 0, 0);
    if (wave > 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) red[wave - 1][e][lane] = acc[e];
    }
    __syncthreads();
    if (wave == 0) {
        // acc[e]: output channel n0 + (e & 3) + 8 (e >> 2) + 4 h, utterance b0 + r
        const int b = b0 + r;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v;
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = ((acc[4 * g + u] + red[0][4 * g + u][lane]) + red[1][4 * g + u][lane]) + red[2][4 * g + u][lane];
            const int n = n0 + 8 * g + 4 * h;
            if (b < B && n + 4 <= N) *reinterpret_cast<f32x4*>(part + ((int64_t)ks * B + b) * N + n) = v;
            else if (b < B) for (int u = 0; u < 4; ++u) if (n + u < N) part[((int64_t)ks * B + b) * N + n + u] = v[u];
        }
    }
}
__global__ __launch_bounds__(256) void rowvec_linear_mfma_sum_kernel(const float* __restrict__ part, int nslices, const float* __restrict__ bias,
                                                                     float* __restrict__ out, int ld_out, int B, int N, int act) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * N) return;
    const int b = i / N, n = i - b * N;
    float s = 0.0f;
    for (int k = 0; k < nslices; ++k) s += part[(int64_t)k * B * N + i];
    out[(int64_t)b * ld_out + n] = apply_act(s + (bias ? bias[n] : 0.0f), act);
}

// ---- finalize the column-sum partials written by the pw2 GEMM epilogue --------------------------------------
// part[((tm*RG + rg)*2 + seg)*C + c]: utterance b owns segment seg = b - (tm*256)/T of tile tm; RG row groups per tile
// (8 x 32 rows from gemm_pw2's LDS image, 2 x 128 rows from gemm_pw3's accumulators).
template <int RG>
__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* __restrict__ part, int64_t sq_stride, int with_std,
                                                              int T, int C, int M, float* __restrict__ out, float eps) {
    const int b = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int r0 = b * T, r1 = r0 + T - 1;
    float s = 0.f, q = 0.f;
    // an utterance of T >= 256 frames touches at most T / 256 + 2 tiles; three at a time so that their 24 (48) loads are all
    // in flight (a runtime-bounded loop exposed one round trip per tile and made this tiny kernel 40 us)
    for (int tm0 = r0 / 256; tm0 <= r1 / 256; tm0 += 3) {
        float ps[3][RG], pq[3][RG];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int tm = tm0 + u;
            const int seg = b - (tm * 256) / T;
            const bool ok = tm <= r1 / 256 && seg >= 0 && seg <= 1;
#pragma unroll
            for (int rg = 0; rg < RG; ++rg) {
                const int64_t o = ((int64_t)(tm * RG + rg) * 2 + (ok ? seg : 0)) * C + c;
                ps[u][rg] = ok ? part[o] : 0.f;
                pq[u][rg] = (ok && with_std) ? part[sq_stride + o] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int rg = 0; rg < RG; ++rg) { s += ps[u][rg]; q += pq[u][rg]; }
    }
    const float mean = s / (float)T;
    if (with_std) {
        out[(int64_t)b * 2 * C + c] = mean;
        out[(int64_t)b * 2 * C + C + c] = sqrtf(fmaxf(q / (float)T - mean * mean, eps));
    } else {
        out[(int64_t)b * C + c] = mean;
    }
    (void)M;
}

// ---- squeeze-excitation MLP: s = sigmoid(W2 relu(W1 m + b1) + b2)  (ECAPA_TDNN.py:171-176) ----------------
// One workgroup per utterance; the hidden layer (H = 128 units) lives in LDS between the two matrix-vector
// products; W1 is [H][C], W2T is the TRANSPOSED second layer [H][C] so that both products read their weights
// coalesced along C in 16-byte loads (the kernel is bound by how many of those each lane keeps in flight:
// 1 MB of fp32 weights per workgroup out of L2).  With `part` the squeeze (mean over frames) is taken straight
// from the column-sum partials of the pw2 GEMM epilogue (layout: colsum_finalize_kernel above).
constexpr int SE_MLP_THREADS = 1024;      // 16 waves: the two matrix-vector products are chains of L2 round trips (256 threads: 33 us)
template <typename WT, int RG>
__global__ __launch_bounds__(SE_MLP_THREADS) void se_mlp_kernel(const float* __restrict__ mean, const float* __restrict__ part, int T,
                                                     const WT* __restrict__ W1, const float* __restrict__ b1,
                                                     const WT* __restrict__ W2T, const float* __restrict__ b2,
                                                     float* __restrict__ s, int B, int C, int H) {
    constexpr int VEC = Vec16<WT>::N;                           // weights per 16-byte load: 4 fp32 / 8 bf16 (bf16 handles)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* xm = reinterpret_cast<float*>(smem);                 // [C]
    f32x4* xm4 = reinterpret_cast<f32x4*>(smem);
    float* hid = xm + C;                                        // [H]
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nch4 = C >> 2, nch = C / VEC;
    if (part) {
        const f32x4* p4 = reinterpret_cast<const f32x4*>(part);
        const int r0 = b * T, r1 = r0 + T - 1;
        const float inv = 1.0f / (float)T;
        for (int ch = threadIdx.x; ch < nch4; ch += SE_MLP_THREADS) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int tm0 = r0 / 256; tm0 <= r1 / 256; tm0 += 3) {       // three tiles' 24 loads in flight (see colsum_finalize_kernel)
                f32x4 ps[3][RG];
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const int tm = tm0 + u;
                    const int seg = b - (tm * 256) / T;
                    const bool ok = tm <= r1 / 256 && seg >= 0 && seg <= 1;
#pragma unroll
                    for (int rg = 0; rg < RG; ++rg)
                        ps[u][rg] = ok ? p4[((int64_t)(tm * RG + rg) * 2 + seg) * nch4 + ch] : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int u = 0; u < 3; ++u)
#pragma unroll
                    for (int rg = 0; rg < RG; ++rg) acc += ps[u][rg];
            }
            xm4[ch] = acc * inv;
        }
    } else {
        const f32x4* m4 = reinterpret_cast<const f32x4*>(mean + (int64_t)b * C);
        for (int ch = threadIdx.x; ch < nch4; ch += SE_MLP_THREADS) xm4[ch] = m4[ch];
    }
    __syncthreads();
    const Vec16<WT>* W1v = reinterpret_cast<const Vec16<WT>*>(W1);
    for (int n0 = wave * 4; n0 < H; n0 += SE_MLP_THREADS / 16) {      // 4 hidden units per wave per pass
        float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int ch = lane; ch < nch; ch += 64) {
            float x[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) x[e] = xm[ch * VEC + e];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const Vec16<WT> w = W1v[(int64_t)(n0 + u) * nch + ch];
#pragma unroll
                for (int e = 0; e < VEC; ++e) a[u] = fmaf(w.get(e), x[e], a[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float t = wave_sum(a[u]);
            if (lane == 0) hid[n0 + u] = fmaxf(t + b1[n0 + u], 0.f);
        }
    }
    __syncthreads();
    // second product: thread = (16-byte chunk of output channels, slice of the hidden units); partial dot products meet in LDS
    // (one thread per chunk walking all H hidden units left 7 of 8 waves idle behind a chain of H / 2 dependent L2 round trips)
    const Vec16<WT>* W2v = reinterpret_cast<const Vec16<WT>*>(W2T);
    float* psum = hid + H;                                      // [slices][C]
    const int slices = SE_MLP_THREADS / nch > 0 ? min(SE_MLP_THREADS / nch, H) : 1;
    const int per = (H + slices - 1) / slices;
    for (int w = threadIdx.x; w < nch * slices; w += SE_MLP_THREADS) {
        const int ch = w % nch, sl = w / nch;
        float a0[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) a0[e] = 0.f;
        const int n1 = min(H, (sl + 1) * per);
#pragma unroll 8
        for (int n = sl * per; n < n1; ++n) {
            const Vec16<WT> w0 = W2v[(int64_t)n * nch + ch];
            const float h0 = hid[n];
#pragma unroll
            for (int e = 0; e < VEC; ++e) a0[e] = fmaf(w0.get(e), h0, a0[e]);
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) psum[sl * C + ch * VEC + e] = a0[e];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += SE_MLP_THREADS) {
        float a = b2[c];
        for (int sl = 0; sl < slices; ++sl) a += psum[sl * C + c];
        s[(int64_t)b * C + c] = 1.0f / (1.0f + expf(-a));
    }
}

// ---- SE gate + residual -------------------------------------------------------------------------
// grid (ceil(T / SE_ROWS), B), block 256: a thread keeps one 16-byte column chunk and walks SE_ROWS frames of ONE utterance,
// so the gate s[b][c..] is loaded once per thread instead of once per output chunk (it was 2x the bytes of the output
// through L1), and 4 frames' loads are in flight per thread.
constexpr int SE_ROWS = 32;
// (fp32 only: `s32` != null also writes the result in the S32 split layout — per row, per 32 channels: 32 hi bf16 | 32 lo bf16 —
//  the A operand format of gemm_pw3's X3 form, so that the next GEMM needs no conversion pass)
template <typename T>
__global__ __launch_bounds__(256) void se_apply_kernel(const T* __restrict__ h, int ldh, const float* __restrict__ s,
                                                       const T* __restrict__ x, int ldx, T* __restrict__ out, int ldo,
                                                       int Tn, int C, char* __restrict__ s32, int ld32,
                                                       const char* __restrict__ x32, int ldx32) {
    // (fp32 instance on F32X3 handles: s32 = the S32 twin of the output; x32 = the residual in the S32 layout instead of x;
    //  out may then be null — every consumer reads the split form, svhip_get_stage converts it back on demand)
    constexpr int VEC = Vec16<T>::N;
    const int cpr = C / VEC;                                  // chunks per row
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * SE_ROWS, t1 = min(t0 + SE_ROWS, Tn);
    for (int cc = threadIdx.x; cc < cpr * 4; cc += 256) {     // 4 frames side by side: thread -> (frame lane fl, chunk)
        const int fl = cc / cpr, c = (cc - fl * cpr) * VEC;
        float g[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) g[j] = s[(int64_t)b * C + c + j];
#pragma unroll 4
        for (int t = t0 + fl; t < t1; t += 4) {
            const int64_t m = (int64_t)b * Tn + t;
            // (h is read exactly once, here: a non-temporal load keeps it from displacing the block input / output in L2 and the MALL)
            Vec16<T> hv;
            hv.v = __builtin_nontemporal_load(reinterpret_cast<const decltype(hv.v)*>(h + m * ldh + c));
            Vec16<T> xv;
            if (VEC == 4 && x32) {
                const char* q = x32 + m * (int64_t)ldx32 * 4 + (c >> 5) * 128 + (c & 31) * 2;
                const x3x4_t xh = *reinterpret_cast<const x3x4_t*>(q), xl = *reinterpret_cast<const x3x4_t*>(q + 64);
#pragma unroll
                for (int j = 0; j < 4; ++j) xv.set(j, static_cast<float>(xh[j]) + static_cast<float>(xl[j]));
            } else {
                xv.v = __builtin_nontemporal_load(reinterpret_cast<const decltype(xv.v)*>(x + m * ldx + c));      // (its last reader before mfa)
            }
            Vec16<T> o;
#pragma unroll
            for (int j = 0; j < VEC; ++j) o.set(j, fmaf(hv.get(j), g[j], xv.get(j)));
            if (out) *reinterpret_cast<Vec16<T>*>(out + m * ldo + c) = o;
            if (VEC == 4 && s32) {
                x3x4_t hi, lo;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = o.get(j);
                    const x3_t hb = x3_hi(v);
                    hi[j] = hb;
                    lo[j] = x3_lo(v, hb);
                }
                char* q = s32 + m * (int64_t)ld32 * 4 + (c >> 5) * 128 + (c & 31) * 2;
                *reinterpret_cast<x3x4_t*>(q) = hi;
                *reinterpret_cast<x3x4_t*>(q + 64) = lo;
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void copy_cols_kernel(const T* __restrict__ src, int lds_, T* __restrict__ dst, int ldd,
                                                        int C, int64_t total_chunks) {
    constexpr int VEC = Vec16<T>::N;
    const int cpr = C / VEC;
    for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total_chunks; id += (int64_t)gridDim.x * 256) {
        const int64_t m = id / cpr;
        const int c = (int)(id - m * cpr) * VEC;
        *reinterpret_cast<Vec16<T>*>(dst + m * ldd + c) = *reinterpret_cast<const Vec16<T>*>(src + m * lds_ + c);
    }
}

// ---- attentive statistics pooling (softmax over T, weighted mean / std, BatchNorm) --------------
// grid (ceil(C / 64), B), block 256: lane = channel, wave w takes frames t = w, w+4, ...
// ONE pass over the logits and X (each 4 B C T per utterance: the kernel is HBM-bound, and a second pass for the variance
// doubled its bytes): online softmax (running max, rescaled sums) with the weighted mean and the weighted sum of squared
// deviations kept in West's incremental form (mean += (w / W) d, M2 += w d (x - mean)): as stable as the two-pass
// form, one exp and one division per element.  The four waves' partial states are merged with the pairwise update.
template <typename T>
__global__ __launch_bounds__(256) void asp_pool_kernel(const float* __restrict__ logits, const T* __restrict__ X, int ldx,
                                                       int Tn, int C, const float* __restrict__ bn_scale,
                                                       const float* __restrict__ bn_shift, float* __restrict__ pooled_raw,
                                                       float* __restrict__ pooled_bn, float eps) {
    __shared__ float smx[4][64], sse[4][64], smean[4][64], sm2[4][64];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const bool ok = c < C;
    const float* __restrict__ lg = logits + (int64_t)b * Tn * C + c;
    const T* __restrict__ xp = X + (int64_t)b * Tn * ldx + c;
    float mx = -INFINITY, se = 0.0f, mean = 0.0f, m2 = 0.0f;
    if (ok) {
        // four frames' loads in flight per lane
        for (int t0 = wave; t0 < Tn; t0 += 16) {
            float a[4], xv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = min(t0 + 4 * u, Tn - 1);
                a[u] = lg[(int64_t)t * C];
                xv[u] = to_f32<T>(xp[(int64_t)t * ldx]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (t0 + 4 * u >= Tn) break;
                if (a[u] > mx) {
                    const float f = expf(mx - a[u]);     // exp(-inf) = 0 on the first frame
                    se *= f;
                    m2 *= f;
                    mx = a[u];
                }
                const float e = expf(a[u] - mx);
                se += e;
                const float d = xv[u] - mean;
                mean = fmaf(e / se, d, mean);
                m2 = fmaf(e * d, xv[u] - mean, m2);
            }
        }
    }
    smx[wave][lane] = mx; sse[wave][lane] = se; smean[wave][lane] = mean; sm2[wave][lane] = m2;
    __syncthreads();
    if (wave == 0 && ok) {
        const float M = fmaxf(fmaxf(smx[0][lane], smx[1][lane]), fmaxf(smx[2][lane], smx[3][lane]));
        float SE = 0.0f, MEAN = 0.0f, M2 = 0.0f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float f = (smx[w][lane] == -INFINITY) ? 0.0f : expf(smx[w][lane] - M);
            const float sw = sse[w][lane] * f;
            if (sw > 0.0f) {
                const float tot = SE + sw;
                const float d = smean[w][lane] - MEAN;
                M2 += sm2[w][lane] * f + d * d * (SE * sw / tot);
                MEAN = fmaf(sw / tot, d, MEAN);
                SE = tot;
            }
        }
        const float sd = sqrtf(fmaxf(M2 / SE, eps));
        if (pooled_raw) {
            pooled_raw[(int64_t)b * 2 * C + c] = MEAN;
            pooled_raw[(int64_t)b * 2 * C + C + c] = sd;
        }
        pooled_bn[(int64_t)b * 2 * C + c] = fmaf(MEAN, bn_scale[c], bn_shift[c]);
        pooled_bn[(int64_t)b * 2 * C + C + c] = fmaf(sd, bn_scale[C + c], bn_shift[C + c]);
    }
}

// ---- eval-mode cropping on device (reference processing/audio_loader.py:110-150 for decoded PCM) -----------------
// pcm: all files' int16 samples back to back; file f = pcm[off[f] .. off[f]+len[f]).  Crop c of file f starts at
// int(linspace(0, n - L, num_eval)[c]) where n = len (or L+1 after wrap padding when len <= L); samples are
// scaled by 1/32768 (soundfile's float32 convention).  One workgroup per crop.
__global__ __launch_bounds__(256) void crop_pcm16_kernel(const int16_t* __restrict__ pcm, const int64_t* __restrict__ off,
                                                         const int32_t* __restrict__ len, int num_eval, int L,
                                                         float* __restrict__ out) {
    const int crop = blockIdx.x;
    const int f = crop / num_eval, c = crop - f * num_eval;
    const int n0 = len[f];
    const int n = n0 <= L ? L + 1 : n0;                       // np.pad(audio, (0, L - n0 + 1), 'wrap')
    // np.linspace(0, n - L, num_eval)[c] = c * ((n - L) / (num_eval - 1)) in float64, truncated by int()
    int start = 0;
    if (num_eval > 1) {
        const double step = (double)(n - L) / (double)(num_eval - 1);
        start = (c == num_eval - 1) ? (n - L) : (int)((double)c * step);
    }
    const int16_t* __restrict__ src = pcm + off[f];
    float* __restrict__ dst = out + (int64_t)crop * L;
    for (int i = threadIdx.x; i < L; i += 256) {
        int j = start + i;
        if (j >= n0) j = (j - n0) % n0;                       // wrap padding reads the file again from its start
        dst[i] = (float)src[j] * (1.0f / 32768.0f);
    }
}

inline int grid_for(int64_t work_items) {
    int64_t g = (work_items + 255) / 256;
    const int64_t cap = 256 * 8;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

hipError_t launch_colmean(const void* X, int dt, int ldx, int B, int T, int C, float* mean, hipStream_t stream,
                          float* scratch, int scratch_slices) {
    const bool bf16 = dt == DT_BF16;
    const int vec = dt != DT_F32 ? 8 : 4;
    if (C % vec != 0 || ldx % vec != 0) return hipErrorInvalidValue;
    const int gx = (C + 64 * vec - 1) / (64 * vec);
    if (scratch && scratch_slices > 1 && T >= 1024 && gx * B < 2048) {       // too few workgroups for a long T: slice it
        const int TS = scratch_slices;
        dim3 grid(gx, B, TS), block(256);
        if (dt == DT_F16) hipLaunchKernelGGL(colsum_slice_kernel<f16_t>, grid, block, 0, stream, (const f16_t*)X, ldx, T, C, TS, scratch);
        else if (bf16) hipLaunchKernelGGL(colsum_slice_kernel<bf16_t>, grid, block, 0, stream, (const bf16_t*)X, ldx, T, C, TS, scratch);
        else hipLaunchKernelGGL(colsum_slice_kernel<float>, grid, block, 0, stream, (const float*)X, ldx, T, C, TS, scratch);
        const int n = B * C;
        hipLaunchKernelGGL(colsum_slice_finalize_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, scratch, TS, C, T, mean, n);
        return hipGetLastError();
    }
    dim3 grid((C + 64 * vec - 1) / (64 * vec), B), block(256);
    if (dt == DT_F16) hipLaunchKernelGGL((colstats_kernel<f16_t, false>), grid, block, 0, stream, (const f16_t*)X, ldx, T, C, mean, C, 0.f);
    else if (bf16) hipLaunchKernelGGL((colstats_kernel<bf16_t, false>), grid, block, 0, stream, (const bf16_t*)X, ldx, T, C, mean, C, 0.f);
    else hipLaunchKernelGGL((colstats_kernel<float, false>), grid, block, 0, stream, (const float*)X, ldx, T, C, mean, C, 0.f);
    return hipGetLastError();
}

hipError_t launch_colstats(const void* X, bool bf16, int ldx, int B, int T, int C, float* stats, float eps, hipStream_t stream) {
    const int vec = bf16 ? 8 : 4;
    if (C % vec != 0 || ldx % vec != 0) return hipErrorInvalidValue;
    dim3 grid((C + 64 * vec - 1) / (64 * vec), B), block(256);
    if (bf16) hipLaunchKernelGGL((colstats_kernel<bf16_t, true>), grid, block, 0, stream, (const bf16_t*)X, ldx, T, C, stats, 2 * C, eps);
    else hipLaunchKernelGGL((colstats_kernel<float, true>), grid, block, 0, stream, (const float*)X, ldx, T, C, stats, 2 * C, eps);
    return hipGetLastError();
}

// ---- the last kernel of every forward: the embeddings leave the workspace, and the numeric status of the call is recorded --------------
// dst[i] = src[i] (dst == src: check only); a value that is not finite — an fp16 activation that overflowed (SVHIP_F16 storage, pack2 ->
// inf), a NaN in the input — raises bit 0 of status[0], is counted in status[1], and sets the host-visible flag word (pinned, mapped
// memory: the host reads it after a stream synchronisation without a copy).  Nothing is written in the normal case.
__global__ __launch_bounds__(256) void emb_out_kernel(const float* __restrict__ src, float* __restrict__ dst, int n, uint32_t* __restrict__ status,
                                                      volatile uint32_t* __restrict__ host_flag) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    bool bad = false;
    if (i < n) {
        const float v = src[i];
        if (dst != src) dst[i] = v;
        bad = !(fabsf(v) <= 3.4028234663852886e38f);         // inf or NaN
    }
    const unsigned long long m = __ballot(bad);
    if (m && (threadIdx.x & 63) == 0) {
        atomicOr(status, 1u);
        atomicAdd(status + 1, (uint32_t)__popcll(m));
        __threadfence_system();
        *host_flag = 1u;
    }
}

hipError_t launch_emb_out(const float* src, float* dst, int n, uint32_t* status, uint32_t* host_flag, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(emb_out_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, src, dst, n, status, host_flag);
    return hipGetLastError();
}

// (short_rows: the caller opts into the 256-wide slices for rows that are not a multiple of 384 — RawNet2's fc on 16-bit handles; the ECAPA
//  layers keep round 4's rule, so that an fp32-grade handle takes the same kernel on either side of B = 64 for K < 3 072)
static int rowvec_linear_slice(int K, bool short_rows) { return (K % 384 == 0 && K >= 8 * 384) ? 384 : (short_rows && K % 256 == 0 && K >= 4 * 256) ? 256 : 0; }
size_t rowvec_linear_scratch_bytes(int B, int N, int K, bool short_rows) { const int ks = rowvec_linear_slice(K, short_rows); return ks ? (size_t)(K / ks) * B * N * sizeof(float) : 0; }

hipError_t launch_rowvec_linear(const float* in, int ld_in, const float* W, const float* bias, float* out, int ld_out,
                                int B, int N, int K, int act, hipStream_t stream, float* part, bool short_rows) {
    if (K <= 0 || N <= 0 || B <= 0 || K % 4 != 0 || ld_in % 4 != 0) return hipErrorInvalidValue;
    if ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(W)) & 15) return hipErrorInvalidValue;
    if (part && B > 64 && N % 4 == 0 && rowvec_linear_scratch_bytes(B, N, K, short_rows) > 0 && (reinterpret_cast<uintptr_t>(part) & 15) == 0) {
        const int ks = rowvec_linear_slice(K, short_rows), ns = K / ks;
        if (ks == 384) hipLaunchKernelGGL(rowvec_linear_mfma_part_kernel<384>, dim3((N + 31) / 32, (B + 31) / 32, ns), dim3(256), 0, stream, in, ld_in, W, part, B, N, K);
        else hipLaunchKernelGGL(rowvec_linear_mfma_part_kernel<256>, dim3((N + 31) / 32, (B + 31) / 32, ns), dim3(256), 0, stream, in, ld_in, W, part, B, N, K);
        hipLaunchKernelGGL(rowvec_linear_mfma_sum_kernel, dim3((B * N + 255) / 256), dim3(256), 0, stream, part, ns, bias, out, ld_out, B, N, act);
        return hipGetLastError();
    }
    if (B <= 64 && K >= 2048) {
        hipLaunchKernelGGL(rowvec_linear_small_kernel, dim3((N + 7) / 8, (B + 3) / 4), dim3(1024), 0, stream, in, ld_in, W, bias, out, ld_out, B, N, K, act);
        return hipGetLastError();
    }
    dim3 grid((N + 7) / 8, (B + 7) / 8), block(256);
    hipLaunchKernelGGL(rowvec_linear_kernel, grid, block, 0, stream, in, ld_in, W, bias, out, ld_out, B, N, K, act);
    return hipGetLastError();
}

// fp32 -> (hi bf16 << 16) | lo bf16 words: the weight-side operand format of gemm_pw's split (F32X3) path
__global__ __launch_bounds__(256) void split_words_kernel(const float* __restrict__ src, uint32_t* __restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = src[i];
        const x3_t h = x3_hi(v);
        const x3_t l = x3_lo(v, h);
        uint16_t hb, lb;
        __builtin_memcpy(&hb, &h, 2);
        __builtin_memcpy(&lb, &l, 2);
        dst[i] = ((uint32_t)hb << 16) | lb;
    }
}

// fp32 rows -> the S32 split layout of gemm_pw3's X3 form: per row, per block of 32 k: 32 hi bf16 | 32 lo bf16 (128 bytes).
// One thread = 8 consecutive k: 32 bytes in, 16 + 16 bytes out.
// `scale` (optional, device: [0] = s, an exact power of two): every value is multiplied by s before it is split (the network input of an
// F32X3 handle: in_scale_kernel below)
__global__ __launch_bounds__(256) void split_s32_kernel(const float* __restrict__ src, int ld, char* __restrict__ dst, int64_t M, int K, int ldd, int kvalid,
                                                        const float* __restrict__ scale) {
    const float sc = scale ? scale[0] : 1.0f;
    const int per_row = K >> 3;
    const int64_t n = M * per_row;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / per_row;
        const int k0 = (int)(i - row * per_row) * 8;
        f32x4 a = {0.0f, 0.0f, 0.0f, 0.0f}, b = a;                   // columns >= kvalid (a multiple of 8) are zero padding
        if (k0 < kvalid) {
            a = *reinterpret_cast<const f32x4*>(src + row * ld + k0);
            b = *reinterpret_cast<const f32x4*>(src + row * ld + k0 + 4);
        }
        x3x8_t hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = (e < 4 ? a[e] : b[e - 4]) * sc;
            const x3_t h = x3_hi(v);
            hi[e] = h;
            lo[e] = x3_lo(v, h);
        }
        char* o = dst + row * (int64_t)ldd * 4 + (k0 >> 5) * 128 + (k0 & 31) * 2;
        *reinterpret_cast<x3x8_t*>(o) = hi;
        *reinterpret_cast<x3x8_t*>(o + 64) = lo;
    }
}

// max |x| over the finite elements of X (256 partial words), then the scale of an F32X3 network input:
//   scale[0] = s, scale[1] = 1 / (s sw), scale[2] = s sw (sw: the weight planes' own scale, ConvLayer::cv_wscale), s = 1 while 2^-8 <= max |x| < 2^13 (the planes carry such values as they are: the arithmetic of rounds 4 - 5,
//   bit for bit), else the power of two that brings max |x| into [64, 128) — features of ANY finite magnitude (log_input = 0: mel power of
//   int16-scaled waveforms is ~1e9) reach the first convolution instead of overflowing its half-precision planes
__global__ __launch_bounds__(256) void absmax_part_kernel(const float* __restrict__ X, int64_t n, uint32_t* __restrict__ part) {
    __shared__ uint32_t wm[4];
    uint32_t m = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = max(m, finite_abs_bits(__float_as_uint(X[i])));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = max(max(wm[0], wm[1]), max(wm[2], wm[3]));
}
__global__ __launch_bounds__(256) void in_scale_kernel(const uint32_t* __restrict__ part, float* __restrict__ scale, float wscale) {
    __shared__ uint32_t wm[4];
    uint32_t m = part[threadIdx.x];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t bits = max(max(wm[0], wm[1]), max(wm[2], wm[3]));
        const int e = (int)(bits >> 23);                      // biased exponent of max |x|
        const float s = (e == 0 || (e >= 127 - 8 && e < 127 + 13)) ? 1.0f : pow2_scale_of_bits(bits);
        // [0]: what the split pass multiplies the input by; [1]: 1 / (s sw), what the GEMM's epilogue multiplies its accumulators by; [2]: s sw,
        // what its accumulators' start value (the bias) is multiplied by.  (s and sw are powers of two: the products and the inverse are exact
        // while they stay inside fp32's exponent range — 2^-122 <= s, sw <= 2^126 each — which the clamp below keeps)
        const float ssw = fminf(fmaxf(s * wscale, 0x1p-120f), 0x1p120f);
        scale[0] = s;
        scale[1] = 1.0f / ssw;
        scale[2] = ssw;
    }
}
hipError_t launch_in_scale(const float* X, int64_t n, uint32_t* part256, float* scale, hipStream_t stream, float wscale) {
    if (!X || !part256 || !scale || n <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(absmax_part_kernel, dim3(256), dim3(256), 0, stream, X, n, part256);
    hipLaunchKernelGGL(in_scale_kernel, dim3(1), dim3(256), 0, stream, part256, scale, wscale);
    return hipGetLastError();
}

hipError_t launch_split_s32(const float* src, int ld, void* dst, int64_t M, int K, hipStream_t stream, int ldd, int kvalid, const float* scale) {
    if (ldd == 0) ldd = K;
    if (kvalid == 0) kvalid = K;
    if (!src || !dst || M <= 0 || K <= 0 || K % 32 != 0 || ld % 4 != 0 || ldd % 32 != 0 || ldd < K || kvalid % 8 != 0 || kvalid > K) return hipErrorInvalidValue;
    const int64_t g = (M * (K / 8) + 255) / 256;
    hipLaunchKernelGGL(split_s32_kernel, dim3((unsigned)(g > 65536 ? 65536 : g)), dim3(256), 0, stream, src, ld, reinterpret_cast<char*>(dst), M, K, ldd, kvalid, scale);
    return hipGetLastError();
}

// S32 -> fp32 (svhip_get_stage on F32X3 handles whose block outputs exist only in the split layout): v = hi + lo
__global__ __launch_bounds__(256) void unsplit_s32_kernel(const char* __restrict__ src, int lds32, float* __restrict__ dst, int ld, int64_t M, int K) {
    const int per_row = K >> 2;
    const int64_t n = M * per_row;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / per_row;
        const int c = (int)(i - row * per_row) * 4;
        const char* q = src + row * (int64_t)lds32 * 4 + (c >> 5) * 128 + (c & 31) * 2;
        const x3x4_t hi = *reinterpret_cast<const x3x4_t*>(q), lo = *reinterpret_cast<const x3x4_t*>(q + 64);
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = static_cast<float>(hi[j]) + static_cast<float>(lo[j]);
        *reinterpret_cast<f32x4*>(dst + row * ld + c) = v;
    }
}

hipError_t launch_unsplit_s32(const void* src, int lds32, float* dst, int ld, int64_t M, int K, hipStream_t stream) {
    if (!src || !dst || M <= 0 || K <= 0 || K % 32 != 0 || ld % 4 != 0 || lds32 % 32 != 0) return hipErrorInvalidValue;
    const int64_t g = (M * (K / 4) + 255) / 256;
    hipLaunchKernelGGL(unsplit_s32_kernel, dim3((unsigned)(g > 65536 ? 65536 : g)), dim3(256), 0, stream, reinterpret_cast<const char*>(src), lds32, dst, ld, M, K);
    return hipGetLastError();
}

hipError_t launch_split_words(const float* src, void* dst, int64_t n, hipStream_t stream) {
    if (!src || !dst || n <= 0) return hipErrorInvalidValue;
    const int64_t g = (n + 255) / 256;
    hipLaunchKernelGGL(split_words_kernel, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, stream, src, reinterpret_cast<uint32_t*>(dst), n);
    return hipGetLastError();
}

hipError_t launch_crop_pcm16(const int16_t* pcm, const int64_t* off, const int32_t* len, int n_files, int num_eval, int L,
                             float* out, hipStream_t stream) {
    if (n_files <= 0 || num_eval <= 0 || L <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(crop_pcm16_kernel, dim3(n_files * num_eval), dim3(256), 0, stream, pcm, off, len, num_eval, L, out);
    return hipGetLastError();
}

hipError_t launch_colsum_finalize(const float* part, int64_t sq_stride, bool with_std, int B, int T, int C, int M,
                                  float* out, float eps, hipStream_t stream, int row_groups) {
    if (row_groups != 8 && row_groups != 2) return hipErrorInvalidValue;
    const dim3 grid((C + 255) / 256, B);
    if (row_groups == 8) hipLaunchKernelGGL(colsum_finalize_kernel<8>, grid, dim3(256), 0, stream, part, sq_stride, with_std ? 1 : 0, T, C, M, out, eps);
    else hipLaunchKernelGGL(colsum_finalize_kernel<2>, grid, dim3(256), 0, stream, part, sq_stride, with_std ? 1 : 0, T, C, M, out, eps);
    return hipGetLastError();
}

hipError_t launch_se_mlp(const float* mean, const float* part, int T, const void* W1, const float* b1, const void* W2T,
                         const float* b2, float* s, bool w_bf16, int B, int C, int H, hipStream_t stream, int row_groups) {
    const int vec = w_bf16 ? 8 : 4;
    if (row_groups != 8 && row_groups != 2) return hipErrorInvalidValue;
    const int slices = std::max(1, std::min(SE_MLP_THREADS / std::max(1, C / vec), H));
    const size_t lds = (size_t)(C + H + (size_t)slices * C) * sizeof(float);
    if (lds > 64 * 1024 || B <= 0 || H % 16 != 0 || C % 8 != 0 || (!mean && !part)) return hipErrorInvalidValue;
#define SVHIP_SE_MLP(WT, RG) hipLaunchKernelGGL((se_mlp_kernel<WT, RG>), dim3(B), dim3(SE_MLP_THREADS), lds, stream, mean, part, T, (const WT*)W1, b1, (const WT*)W2T, b2, s, B, C, H)
    if (w_bf16) { if (row_groups == 8) SVHIP_SE_MLP(bf16_t, 8); else SVHIP_SE_MLP(bf16_t, 2); }
    else { if (row_groups == 8) SVHIP_SE_MLP(float, 8); else SVHIP_SE_MLP(float, 2); }
#undef SVHIP_SE_MLP
    return hipGetLastError();
}

hipError_t launch_se_apply(const void* h, int ldh, const float* s, const void* x, int ldx, void* out, int ldo,
                           bool bf16, int B, int T, int C, hipStream_t stream, void* s32, int ld32, const void* x32, int ldx32) {
    const int vec = bf16 ? 8 : 4;
    if (C % vec || ldh % vec || ldx % vec || ldo % vec || B <= 0 || T <= 0) return hipErrorInvalidValue;
    if (s32 && (bf16 || C % 32 != 0 || ld32 % 32 != 0 || (reinterpret_cast<uintptr_t>(s32) & 127))) return hipErrorInvalidValue;
    if (x32 && (bf16 || C % 32 != 0 || ldx32 % 32 != 0 || (reinterpret_cast<uintptr_t>(x32) & 127))) return hipErrorInvalidValue;
    if ((!x && !x32) || (!out && !s32)) return hipErrorInvalidValue;
    dim3 grid((T + SE_ROWS - 1) / SE_ROWS, B), block(256);
    if (bf16) hipLaunchKernelGGL(se_apply_kernel<bf16_t>, grid, block, 0, stream, (const bf16_t*)h, ldh, s, (const bf16_t*)x, ldx, (bf16_t*)out, ldo, T, C, (char*)nullptr, 0, (const char*)nullptr, 0);
    else hipLaunchKernelGGL(se_apply_kernel<float>, grid, block, 0, stream, (const float*)h, ldh, s, (const float*)x, ldx, (float*)out, ldo, T, C, (char*)s32, ld32, (const char*)x32, ldx32);
    return hipGetLastError();
}

hipError_t launch_copy_cols(const void* src, int lds_, void* dst, int ldd, bool bf16, int M, int C, hipStream_t stream) {
    const int vec = bf16 ? 8 : 4;
    if (C % vec || lds_ % vec || ldd % vec) return hipErrorInvalidValue;
    const int64_t chunks = (int64_t)M * (C / vec);
    dim3 grid(grid_for(chunks)), block(256);
    if (bf16) hipLaunchKernelGGL(copy_cols_kernel<bf16_t>, grid, block, 0, stream, (const bf16_t*)src, lds_, (bf16_t*)dst, ldd, C, chunks);
    else hipLaunchKernelGGL(copy_cols_kernel<float>, grid, block, 0, stream, (const float*)src, lds_, (float*)dst, ldd, C, chunks);
    return hipGetLastError();
}

hipError_t launch_asp_pool(const float* logits, const void* X, bool bf16, int ldx, int B, int T, int C,
                           const float* bn_scale, const float* bn_shift, float* pooled_raw, float* pooled_bn,
                           float eps, hipStream_t stream) {
    dim3 grid((C + 63) / 64, B), block(256);
    if (bf16) hipLaunchKernelGGL(asp_pool_kernel<bf16_t>, grid, block, 0, stream, logits, (const bf16_t*)X, ldx, T, C, bn_scale, bn_shift, pooled_raw, pooled_bn, eps);
    else hipLaunchKernelGGL(asp_pool_kernel<float>, grid, block, 0, stream, logits, (const float*)X, ldx, T, C, bn_scale, bn_shift, pooled_raw, pooled_bn, eps);
    return hipGetLastError();
}

}  // namespace svhip
