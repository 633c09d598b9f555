// kernels.h — internal launcher API of libsvhip (host side). All launchers enqueue on `stream`.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <functional>

#include "common.h"

struct svhip_handle;

namespace svhip {

// ---------------------------------------------------------------------------------------------
// Handle accessors for the translation units that do not see the handle's definition (api.hip owns it)
// ---------------------------------------------------------------------------------------------
hipStream_t handle_stream(svhip_handle* h);
int handle_device(const svhip_handle* h);
void handle_set_error(svhip_handle* h, const char* msg);
void*& handle_comm(svhip_handle* h);            // opaque slot owned by comm.hip
// profiling-aware launch (the same event bracketing api.hip's own launches get); returns an svhip_status
int handle_run(svhip_handle* h, const char* label, const std::function<hipError_t()>& launch);

// ---------------------------------------------------------------------------------------------
// One-time per-DEVICE setup flag.  hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the function on the
// CURRENT device only, so a launcher that raises a kernel's LDS limit keeps one bit per device ordinal (a process may
// own handles on several GPUs).  Lock-free: two threads racing on the same device both set the (idempotent) attribute.
// ---------------------------------------------------------------------------------------------
struct DeviceOnce {
    std::atomic<uint64_t> mask{0};
    bool done(int dev) const { return dev >= 0 && dev < 64 && ((mask.load(std::memory_order_acquire) >> dev) & 1u); }
    void mark(int dev) { if (dev >= 0 && dev < 64) mask.fetch_or(1ull << dev, std::memory_order_release); }
};
inline hipError_t set_max_dynamic_lds(DeviceOnce& once, const void* fn, int bytes) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (once.done(dev)) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) once.mark(dev);
    return e;
}

// ---------------------------------------------------------------------------------------------
// Tiled MFMA GEMM with fused conv-gather prologue and bias/activation/BN epilogue.
//   Y[m, n] = act2( scale[n] * act1( sum_k A'[m, k] * W[n, k] + bias[n] (+ bias_utt[m / T, n]) ) + shift[n] )
// A' is A (frame-major, row stride lda) or, with taps > 1, the im2col view of a dilated 1-D
// convolution over the T rows of each utterance: k = tap*cin + c reads row t + (tap - taps/2)*dil
// (reflect or zero padded).  With A2 != nullptr the operand is A + A2 (Res2Net chain).
// W is packed [Np][Kp] (K contiguous; zero padded to the tile sizes) in the compute dtype.
// ---------------------------------------------------------------------------------------------
struct GemmParams {
    const void* A = nullptr;
    const void* A2 = nullptr;
    const void* W = nullptr;
    void* Y = nullptr;
    const float* bias = nullptr;
    const float* bias_utt = nullptr;
    const float* scale = nullptr;
    const float* shift = nullptr;
    // pw2 / pw3 only: per-utterance column sums of the output (pw2: of the bf16-rounded LDS output tile, 8 row groups of 32 rows
    // per tile; pw3: of the activated fp32 accumulators, 2 row groups of 128 rows — gemm_colsum_groups says which):
    //   colsum[((tile_m*RG + rg)*2 + seg) * N + n] = sum over the rows of row group rg that belong to
    //   utterance (tile_m*256 / T + seg); with colsum_sq the sums of squares follow at offset colsum_stride.
    float* colsum = nullptr;
    int colsum_sq = 0;
    int64_t colsum_stride = 0;
    const float* zeros = nullptr;    // >= N zero floats / >= N ones: stand-ins for absent bias / scale / shift vectors in the kernels that
    const float* ones = nullptr;     // read all three by DMA (gemm_pw3's 16-bit conv-gather form)
    const void* zero_page = nullptr; // >= 64 zero bytes (LDS-DMA source for padded k / out-of-range frames in the conv-gather pw2 path)
    void* Y2 = nullptr;             // gemm_pw3's Res2Net step form only: second output (the next step's input), row stride lda2
    const void* R = nullptr;        // optional residual (M, ldr) in the activation dtype, added last
    int ldr = 0;
    // conv-gather GEMMs on the 256 x 256 bf16 kernel only: K3 extra K columns appended after the taps * cin conv columns, read
    // from row m of a second matrix (the 1 x 1 shortcut of a RawNet2 block as part of conv2's GEMM: W is [N][K + K3], Kp = K + K3)
    const void* A3 = nullptr;
    int lda3 = 0, K3 = 0;
    // x3 == 2, pointwise form only: channels [0, side_c) of the output go to side_a and [side_c, 2 side_c) to side_b in the S32 split
    // layout (row strides in elements) INSTEAD of the fp32 output — the Res2Net pass-through chunk and the first step's input of
    // an F32X3 handle (side_c = C / 8 = 64 or 128)
    int y_s32 = 0;                  // x3 == 2, pointwise / conv-gather forms: Y is written in the S32 split layout (ldy in elements) instead of fp32
    void* side_a = nullptr;
    void* side_b = nullptr;
    int side_lda = 0, side_ldb = 0, side_c = 0;
    int num_cu = 256;               // compute units of the device (grid-size routing, gemm_route)
    int n128_off = 0;               // developer option n128_off: N = 128 layers stay on gemm_pw
    int cv_off = 0;                 // developer option cv_off: 16-bit conv-gather GEMMs stay on the per-tile kernel
    int pw3_cus = -1;               // developer option pw3_cus (svhip_set_option / SVHIP_PW3_CUS at create): the persistent kernels launch at most this
                                    // many workgroups, so that a small test problem walks several tiles per workgroup; 0: persistent kernels off
    const float* in_scale = nullptr; // gemm_pw3's X3 conv-gather form: (device) [0] = s, [1] = 1 / (s sw), [2] = s sw — A holds s * x, W holds sw * w (launch_in_scale):
                                    // the accumulators start at s sw * bias and the epilogue multiplies by 1 / (s sw) before the activation
    int pw4 = 0;                    // option pw4: plain pointwise bf16 layers with more tiles than workgroups run on the four-wave kernel (gemm_pw4.hip)
    int tail_split = 1;             // persistent 16-bit GEMMs: a last partial round of <= G / 2 tiles is walked as column halves (gemm_pw3.hip)
    void* ts = nullptr;             // developer builds (SVHIP_GEMM_DEBUG, debug bit 16384): per-workgroup stage timestamps
    int M = 0, N = 0, K = 0, Kp = 0;
    int lda = 0, lda2 = 0, ldy = 0, ld_bu = 0;
    int T = 1;
    uint32_t t_magic = 0;     // gemm_pw3's gather forms: ceil(2^32 / T), filled by their launchers (frame index = row mod T without a division)
    int taps = 1, dil = 1, cin = 0, pad_mode = 0;
    int act1 = 0, act2 = 0;
    int out_f32 = 0;          // bf16 compute only: store fp32 instead of bf16
    int f16 = 0;              // 16-bit compute only: operands and outputs are fp16 (SVHIP_F16 handles), not bf16
    int x3 = 0;               // fp32 operands only: products as three bf16 MFMAs on hi / lo-split fragments.  1: gemm_pw expects W as
                              // (hi bf16 << 16 | lo bf16) words (ConvLayer::Wsplit), the generic kernel splits true fp32 W itself;
                              // 2: A and W are both in the S32 split layout (gemm_pw3's X3 form, launch_gemm_pw3x3)
    int debug = 0;            // developer ablations (tools/gemm_bench): 1 no loads in the loop, 2 no MFMA, 4 no epilogue
    int Wrows = 0;            // allocated rows of W (loads clamp to Wrows-1); packed weights: N rounded up to 128
};

constexpr int GEMM_BM = 128;
constexpr int GEMM_BN = 128;
constexpr int GEMM_BK_BYTES = 128;   // K-step = 128 bytes per row: 32 fp32 or 64 bf16

inline int gemm_bk(bool bf16) { return bf16 ? 64 : 32; }
inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

hipError_t launch_gemm(const GemmParams& p, bool bf16, hipStream_t stream);
// pointwise fast path (gemm_pw.hip): 256 x 128 tiles, LDS-DMA ring; launch_gemm routes to it when supported
// Which kernel launch_gemm runs for a shape (the profile labels of api.hip name the same choice):
//   PW2        256 x 256 role-staggered bf16 kernel (gemm_pw2.hip)
//   PW_NARROW  gemm_pw's 256 x 128 tile: a pw2 grid of at most half the CUs finishes in one round either way, and the half-size
//              tile takes about half as long (RawNet2 blocks 6 / 7, 86 tiles: 370 -> 470 - 510 TFLOP/s)
//   PW         gemm_pw with its default tile;  GENERIC  the register-staged kernel of gemm.hip
//   PW3        the persistent form of PW2 with the epilogue taken from the accumulators (gemm_pw3.hip): plain pointwise layers
//              with more tiles than CUs (ECAPA's tdnn1 / tdnn2 / mfa)
//   PW3CV      the persistent kernel's conv-gather form on 16-bit operands: k = 3 / 5 convolutions with cin % 64 == 0 and more tiles
//              than CUs (RawNet2 blocks 2 - 5, ECAPA blocks.0 with padded input rows)
//   N128       bf16 pointwise layers with N = 128 and the ReLU -> BN -> tanh epilogue (ECAPA's asp.tdnn): 128 x 128 tiles, two workgroups
//              per CU (gemm_n128.hip), every batch size
enum GemmRoute : int { ROUTE_PW2 = 0, ROUTE_PW = 1, ROUTE_PW_NARROW = 2, ROUTE_GENERIC = 3, ROUTE_PW3 = 4, ROUTE_PW3CV = 5, ROUTE_N128 = 6 };
GemmRoute gemm_route(const GemmParams& p, bool bf16);
bool gemm_pw_supported(const GemmParams& p, bool bf16);
hipError_t launch_gemm_pw(const GemmParams& p, bool bf16, hipStream_t stream, bool narrow = false);
// bf16 256 x 256 role-staggered variant for the big layers (gemm_pw2.hip); launch_gemm_pw routes to it
bool gemm_pw2_supported(const GemmParams& p, bool bf16);
hipError_t launch_gemm_pw2(const GemmParams& p, hipStream_t stream);
// persistent variant (gemm_pw3.hip); pw3_grid_cap: workgroups it launches at most (the CU count, or the developer override
// SVHIP_PW3_CUS, which lets small test problems run more than one tile per workgroup)
bool gemm_pw3_supported(const GemmParams& p, bool bf16);
hipError_t launch_gemm_pw3(const GemmParams& p, hipStream_t stream);
int pw3_grid_cap(const GemmParams& p);
// round 6: the same contract on four waves with 128 x 128 wave tiles (gemm_pw4.hip)
bool gemm_pw4_supported(const GemmParams& p, bool bf16);
hipError_t launch_gemm_pw4(const GemmParams& p, hipStream_t stream);
// the X3 form of the persistent kernel (x3 == 2): A (M, K) and W (N, K) in the S32 split layout (per row, per 32 k: 32 hi bf16 |
// 32 lo bf16), fp32 output, exact GELU + BN affine, optional column sums: the GELU layers of SVHIP_F32X3 handles
bool gemm_pw3x3_supported(const GemmParams& p);
hipError_t launch_gemm_pw3x3(const GemmParams& p, hipStream_t stream);
bool r2_step_supported(const GemmParams& p);               // r2_step.hip: the Res2Net step on 128 x 128 tiles, two workgroups per CU (cin = 128)
hipError_t launch_r2_step(const GemmParams& p, hipStream_t stream);
// RawNet2's 128 -> 128, k = 3 convolutions on F32X3 handles (r2_step.hip, modes 1 / 2): S32 operands, zero padding (GemmParams::zero_page),
// mode 1: S32 output of lrelu0.3(BN(.)); mode 2: fp32 output of conv (+ R)
bool rn_step_supported(const GemmParams& p, int mode);
hipError_t launch_rn_step(const GemmParams& p, int mode, hipStream_t stream);
bool gemm_pw3cv_supported(const GemmParams& p);          // conv-gather X3 form (odd taps, reflect): see gemm_pw3.hip
hipError_t launch_gemm_pw3cv(const GemmParams& p, hipStream_t stream);
bool gemm_n128_supported(const GemmParams& p);           // gemm_n128.hip
hipError_t launch_gemm_n128(const GemmParams& p, hipStream_t stream);
bool gemm_pw3cv16_supported(const GemmParams& p);        // conv-gather form on bf16 / fp16 operands (reflect or zero padding)
hipError_t launch_gemm_pw3cv16(const GemmParams& p, hipStream_t stream);
// one Res2Net step of an F32X3 handle on the same kernel (dilated k = 3 conv gathered from an S32 input, outputs in S32)
bool gemm_pw3r2_supported(const GemmParams& p);
hipError_t launch_gemm_pw3r2(const GemmParams& p, hipStream_t stream);
// fp32 (M, K) rows (stride ld) -> S32 layout, rows of ldd elements (4 bytes each; 0: dense, ldd = K)
hipError_t launch_split_s32(const float* src, int ld, void* dst, int64_t M, int K, hipStream_t stream, int ldd = 0, int kvalid = 0,
                            const float* scale = nullptr);      // scale (device, optional): [0] = a power of two every value is multiplied by
// the input scale of an F32X3 handle's first convolution: scale[0] = s, scale[1] = 1 / s from the max |x| of X (elementwise.hip)
hipError_t launch_in_scale(const float* X, int64_t n, uint32_t* part256, float* scale, hipStream_t stream, float wscale = 1.0f);
hipError_t launch_unsplit_s32(const void* src, int lds32, float* dst, int ld, int64_t M, int K, hipStream_t stream);   // v = hi + lo
// row groups per 256-row tile in the column-sum partials the routed kernel writes (8: pw2, 2: pw3)
int gemm_colsum_groups(const GemmParams& p, bool bf16);

// ---------------------------------------------------------------------------------------------
// Fused Res2Net chain (bf16 path): one workgroup per utterance runs the 7 dependent dilated convs
// of one SE-Res2Net block with the conv input resident in LDS (res2net.hip).
// H1 = tdnn1 output (M, ld), H2 = chain output (M, ld): y_0 = c_0, y_j = BN(ReLU(conv(c_j + y_{j-1}))).
// ---------------------------------------------------------------------------------------------
struct Res2Params {
    const void* H1 = nullptr;
    void* H2 = nullptr;
    int ld = 0, T = 0, dil = 1, Kp = 0;
    const void* W[7] = {};       // packed [>= C/8 rows][Kp] bf16, k = tap * C/8 + c
    const float* bias[7] = {};
    const float* scale[7] = {};
    const float* shift[7] = {};
    int debug = 0;               // tools/res2_bench ablations (only read in -DSVHIP_GEMM_DEBUG builds)
    unsigned long long* ts = nullptr;   // the same builds, debug bit 64: per workgroup [8 stages][4] cycle totals {taps, epilogue, row pass, -}
    int slices = 1;              // > 1: every utterance is cut into this many time slices, one workgroup each (res2net_chain_slices)
    int Tc = 0;                  // filled by the launcher: core frames per slice
};
bool res2net_chain_supported(int C, int T, int dil, int Kp);
int res2net_chain_slices(int B, int C, int T, int dil, int num_cu);
hipError_t launch_res2net_chain(const Res2Params& p, int B, int C, hipStream_t stream);

// ---------------------------------------------------------------------------------------------
// Mel front-end
// ---------------------------------------------------------------------------------------------
struct FbankTables {           // device pointers, built once per handle
    const float* basis = nullptr;      // [q][tile][lane] float4: windowed cos/sin taps laid out for MFMA B operands
    const void* basis_hi = nullptr;    // bf16x3 path: [k16][pair][part][lane] 8 x bf16 (hi / lo parts of the same taps)
    const void* basis_lo = nullptr;
    const void* basis_l3 = nullptr;    // bf16x6 path (F32X3 handles): the third part of the exact split basis = hi + lo + l3
    const void* sym_hi = nullptr;      // fused front-end (bf16 handles): [7 k steps][8 bin groups][cos | sin][lane] 8 x bf16 of the symmetric basis
    const void* sym_lo = nullptr;      //   w_m cos(2 pi k m / 512) | w_m sin(2 pi k m / 512), m = 0 .. 100 (fbank.hip, "the fused front-end")
    int split6 = 0;
    int n_k16 = 13;                    // ceil(win_length / 16)
    int split_bf16 = 0;                // 1: bf16x3 DFT (bf16-compute handles)
    const float* mel_w = nullptr;      // packed non-zero mel weights
    int n_melw = 0;                    // number of packed weights
    const int* mel_start = nullptr;    // [n_mels] first bin
    const int* mel_len = nullptr;      // [n_mels] number of bins
    const int* mel_off = nullptr;      // [n_mels] offset into mel_w
    int n_fft = 512, win_length = 200, hop = 80, n_mels = 80, n_bins = 257, lpad = 156;
    int n_pairs = 9;                   // ceil(n_bins / 32) re/im tile pairs
    int n_q = 25;                      // win_length / 8
    int mel_max_bin = 256;             // highest bin with a non-zero mel weight
    int force32 = 0;                   // developer option fbank32: the 32-frame kernel whatever the bank
    int ff_abl = 0;                    // developer option ff_abl (tools only; results are then wrong): fused front-end phases skipped: 1 sample loads,
                                       // 2 operand build, 4 MFMAs, 8 mel / log, 16 row stores, 32 the normalisation launch
    float preemph = 0.97f;
};
// wav (B, L) fp32 -> mel power (B, n_mels, T) fp32
hipError_t launch_fbank(const FbankTables& tb, const float* wav, int B, int L, int T, float* mel, hipStream_t stream);

// round 6, bf16 handles: wav (B, L) -> the 16-bit frame-major operand of blocks.0 (log-mel minus its time mean) in two launches
bool fbank_fused_supported(const FbankTables& tb, int L);
hipError_t launch_fbank_fused(const FbankTables& tb, const float* wav, int B, int L, int T, int log_input, float* logmel, float* partial,
                              void* out, hipStream_t stream);

// (B, n_mels, T) fp32 features -> frame-major (B, T, n_mels) activations in the compute dtype,
// with optional log(x+1e-6) - mean_t and optional InstanceNorm1d(affine).
// `stats` is a (B * n_mels * 2) fp32 scratch buffer.
hipError_t launch_prologue(const float* feat, void* out, bool out_bf16, int B, int n_mels, int T,
                           int log_input, const float* in_w, const float* in_b, float* stats, hipStream_t stream,
                           uint32_t* status = nullptr, uint32_t* host_flag = nullptr, float limit = 3.0e38f);       // range guard: fbank.hip

// ---------------------------------------------------------------------------------------------
// Frame-major reductions / elementwise (activation dtype templated inside)
// ---------------------------------------------------------------------------------------------
// mean over the T rows of each utterance: X (B*T, ldx) cols [0,C) -> mean (B, C) fp32
// (optional scratch of B * scratch_slices * C floats: long T with few channels is reduced in two stages)
hipError_t launch_colmean(const void* X, int dt, int ldx, int B, int T, int C, float* mean, hipStream_t stream,
                          float* scratch = nullptr, int scratch_slices = 0);
// mean and population std over T (two pass, clamp 1e-12 as ECAPA_TDNN.py:222-227): -> stats (B, 2C) = [mean | std]
hipError_t launch_colstats(const void* X, bool bf16, int ldx, int B, int T, int C, float* stats, float eps, hipStream_t stream);
// out[b, n] = act( bias[n] + sum_k W[n, k] * in[b, k] ), all fp32 (small-M linear layers)
// `part` (optional, rowvec_linear_scratch_bytes(B, N, K) > 0 bytes): full batches (B > 64) of long rows (K a multiple of 384, >= 3 072) run on the
// exact fp32 MFMA with K split over workgroups and the slices added in a fixed order
// `short_rows`: also rows that are a multiple of 256 (>= 1 024) but not of 384, on slices of 256 (RawNet2's fc, 16-bit handles)
size_t rowvec_linear_scratch_bytes(int B, int N, int K, bool short_rows = false);
hipError_t launch_rowvec_linear(const float* in, int ld_in, const float* W, const float* bias, float* out, int ld_out,
                                int B, int N, int K, int act, hipStream_t stream, float* part = nullptr, bool short_rows = false);
// the embeddings leave the workspace (dst == src: no copy) and the call's numeric status is recorded: a non-finite value raises bit 0 of
// status[0], is counted in status[1] and sets *host_flag (mapped pinned memory).  SVHIP_STATUS_* bits: api.hip
hipError_t launch_emb_out(const float* src, float* dst, int n, uint32_t* status, uint32_t* host_flag, hipStream_t stream);
// s[b, :] = sigmoid(W2 relu(W1 mean[b, :] + b1) + b2); W1 [H][C], W2T = W2 transposed [H][C]
hipError_t launch_se_mlp(const float* mean, const float* part, int T, const void* W1, const float* b1, const void* W2T,
                         const float* b2, float* s, bool w_bf16, int B, int C, int H, hipStream_t stream, int row_groups = 8);
// out[(b,t), c] = h[(b,t), c] * s[b, c] + x[(b,t), c]   (SE gate + residual, ECAPA_TDNN.py:177,336)
// (fp32: s32 != null also writes out in the S32 split layout, row stride ld32 elements of 4 bytes: gemm_pw3's X3 A operand)
hipError_t launch_se_apply(const void* h, int ldh, const float* s, const void* x, int ldx, void* out, int ldo,
                           bool bf16, int B, int T, int C, hipStream_t stream, void* s32 = nullptr, int ld32 = 0, const void* x32 = nullptr, int ldx32 = 0);
// reduce the pw2 column-sum partials: out (B, C) = mean over T  [and out (B, 2C) = [mean | std] with sq]
hipError_t launch_colsum_finalize(const float* part, int64_t sq_stride, bool with_std, int B, int T, int C, int M,
                                  float* out, float eps, hipStream_t stream, int row_groups = 8);
// fp32 matrix -> (hi bf16 << 16 | lo bf16) words (the pre-split weight operand of gemm_pw's F32X3 path)
hipError_t launch_split_words(const float* src, void* dst, int64_t n, hipStream_t stream);
// eval-mode crops of int16 PCM files (back to back in `pcm`) -> (n_files * num_eval, L) fp32, 1/32768 scaling
hipError_t launch_crop_pcm16(const int16_t* pcm, const int64_t* off, const int32_t* len, int n_files, int num_eval, int L,
                             float* out, hipStream_t stream);
// strided 2-D copy of a column block: dst[m, 0:C) = src[m, 0:C)
hipError_t launch_copy_cols(const void* src, int lds, void* dst, int ldd, bool bf16, int M, int C, hipStream_t stream);
// attentive statistics: softmax over T of logits (fp32, ld = C), weighted mean / std of X, then
// BatchNorm affine over the 2C pooled values -> pooled (B, 2C) fp32   (ECAPA_TDNN.py:252-259,496)
hipError_t launch_asp_pool(const float* logits, const void* X, bool bf16, int ldx, int B, int T, int C,
                           const float* bn_scale, const float* bn_shift, float* pooled_raw, float* pooled_bn,
                           float eps, hipStream_t stream);

// Fused attention tail (bf16 path, asp_fused.hip): logits = conv1x1(att) + b, softmax over T, weighted
// mean / std of X, BatchNorm affine -> pooled (B, 2C).  The logits never reach memory.
struct AspFusedParams {
    const void* att = nullptr;      // (B*T, 128) bf16
    const void* W = nullptr;        // asp.conv packed [>= C rows][Kp = 128] bf16
    const float* bias = nullptr;    // [C]
    const void* X = nullptr;        // (B*T, ldx) bf16, the mfa output
    const float* bn_scale = nullptr;
    const float* bn_shift = nullptr;
    float* pooled_raw = nullptr;    // optional (B, 2C)
    float* pooled_bn = nullptr;     // (B, 2C)
    int ldx = 0, T = 0, C = 0, Kp = 0;
    float eps = 1e-12f;
    unsigned long long* dbg = nullptr;   // tools/asp_bench: per-phase s_memtime totals of wave 0 (only read in -DSVHIP_GEMM_DEBUG builds)
};
bool asp_fused_supported(int T, int C, int att_channels, int Kp);
hipError_t launch_asp_fused(const AspFusedParams& p, int B, hipStream_t stream);

// the same fusion for SVHIP_F32X3 handles (asp_x3.hip): fp32 att and x, logits as split-bf16 MFMA triples, never stored
struct AspX3Params {
    const float* att = nullptr;     // (B*T, 128) fp32, asp.tdnn's output
    const void* Ws32 = nullptr;     // asp.conv in the S32 split layout: [C rows][128 k]: per 32 k, 32 hi | 32 lo bf16
    const float* X = nullptr;       // (B*T, ldx) fp32, the mfa output
    const float* mref = nullptr;    // (B, mref_ld): the plain mean over time of every channel of X (the shift of the moments)
    const float* bn_scale = nullptr;
    const float* bn_shift = nullptr;
    float* pooled_raw = nullptr;    // optional (B, 2C)
    float* pooled_bn = nullptr;     // (B, 2C)
    int ldx = 0, mref_ld = 0, T = 0, C = 0, B = 0;
    float eps = 1e-12f;
};
hipError_t launch_asp_bf16(const AspFusedParams& p, const float* mref, int mref_ld, int B, hipStream_t stream);      // asp_x3.hip: the bf16 form of asp_x3
bool asp_x3_supported(int T, int C, int att_channels, int K);
hipError_t launch_asp_x3(const AspX3Params& p, int B, hipStream_t stream);

// ---------------------------------------------------------------------------------------------
// RawNet2 (rawnet2.hip)
// ---------------------------------------------------------------------------------------------
// (xn: optional bf16 copies of the LayerNorm output, per utterance two zero-tailed rows of Lp >= L + RN_XN_TAIL samples (Lp % 64
//  == 0): sample j at index j, then sample j + 1 at index j — the operand of the bf16 sinc kernel, which reads up to 470 samples
//  past its last tile's first sample)
constexpr int RN_XN_TAIL = 512;
// (xn_lo: the split form of F32X3 handles — FOUR rows of Lp halves per utterance: hi parts of the two copies, then their lo parts)
hipError_t launch_rn_ln_stats(const float* wav, int B, int L, float* stats, hipStream_t stream, void* xn = nullptr, int Lp = 0,
                              const float* gamma = nullptr, const float* beta = nullptr, int xn_dt = DT_BF16, bool xn_lo = false);
// the sinc front-end on three fp16 MFMAs per product (F32X3 handles): filt_planes = [2][128][256] halves (hi | lo), fp32 out
// (pre_s32: optional second output lrelu0.3(next_scale * out + next_shift) in the S32 split layout, (B * T1, 128))
hipError_t launch_rn_sinc_x3(const void* filt_planes, const float* bn_scale, const float* bn_shift, float* out, int B, int L, int T1,
                             const void* xn, int Lp, int num_cu, hipStream_t stream, void* pre_s32 = nullptr, const float* next_scale = nullptr,
                             const float* next_shift = nullptr);
// LayerNorm + sinc conv (k=251) + abs + maxpool3 + BN + LeakyReLU(0.3): wav (B, L) -> out (B, T1, 128), T1 = (L-250)/3
hipError_t launch_rn_sinc(const float* wav, const float* stats, const float* gamma, const float* beta, const void* filt,
                          const float* bn_scale, const float* bn_shift, void* out, int dt, int B, int L, int T1,
                          hipStream_t stream, void* pre = nullptr, const float* next_scale = nullptr, const float* next_shift = nullptr,
                          const void* xn = nullptr, int Lp = 0, int num_cu = 256, bool sym = false);
// (sym, DT_F16 only: `filt` is the [128][128] slot-major table of the symmetric form — rawnet2.hip, rn_sinc_kernel<.., SYM>)
// (dt: DT_F32 / DT_BF16 / DT_F16 — the storage type of the activations)
// (y_s32 / pre_s32, fp32 only: that output in the S32 split layout — the operand of the split convolution kernels — instead of fp32)
hipError_t launch_rn_bn_act(const void* x, void* y, int dt, const float* scale, const float* shift, int64_t rows, int C,
                            float slope, hipStream_t stream, bool y_s32 = false);
hipError_t launch_rn_maxpool3(const void* x, void* y, int dt, int B, int Tin, int C, hipStream_t stream);
hipError_t launch_rn_afms_apply(const void* x, void* y, int dt, const float* alpha, const float* s, int B, int T, int C,
                                hipStream_t stream, const float* next_scale = nullptr, const float* next_shift = nullptr,
                                void* pre = nullptr, float slope = 0.3f, bool pre_s32 = false);
// fused block tail (rawnet2.hip): [max_pool1d(3)] + AFMS + the next consumer's lrelu(bn(.)), one workgroup per utterance with the
// pooled activation held in registers; rn_tail_supported says whether (Tn, C) fits (else the four separate passes run)
bool rn_tail_supported(int dt, int Tn, int C);
hipError_t launch_rn_tail(const void* x, void* y, void* pre, int dt, bool pool, const float* alpha, const float* WT, const float* bias,
                          const float* next_scale, const float* next_shift, int B, int Tin, int C, float slope, hipStream_t stream,
                          const void* res = nullptr, float* part = nullptr, float* gate = nullptr, int num_cu = 0, bool pre_s32 = false);
// small batches (B * 4 <= num_cu) with `part` (B x 16 x C floats) and `gate` (B x C floats): the utterance in frame slices over several
// workgroups (slice sums -> rn_afms_gate -> apply); returns the slice count, 0 = the one-workgroup-per-utterance kernel
int rn_tail_slices(int dt, int B, int Tn, int C, int num_cu);       // res: the block input, added to x before the pool (16-bit handles; see rn_tail_kernel)
// Fused 128 -> 128 pooled RawNetBasicBlock (rn_block128.hip, bf16): previous block's AFMS gate on the way in, BN + LeakyReLU,
// conv1 + BN + LeakyReLU, conv2 + identity shortcut, max_pool1d(3), per-tile column sums of the pooled output.
struct RnBlock128Params {
    const bf16_t* xin = nullptr;         // (B, T, 128): the stage input before the previous block's gate
    const float* alpha = nullptr;        // [128] previous block's AFMS alpha, or null (first block)
    const float* gate = nullptr;         // (B, 128) previous block's AFMS sigmoid(fc(mean)), or null
    const float* bn1_scale = nullptr;    // [128] folded bn1
    const float* bn1_shift = nullptr;
    const bf16_t* W1 = nullptr;          // packed [128][384], k = tap * 128 + c
    const float* bn2_scale = nullptr;    // [128] folded bn2 (conv1's epilogue)
    const float* bn2_shift = nullptr;
    const bf16_t* W2 = nullptr;
    bf16_t* opool = nullptr;             // (B, T / 3, 128)
    float* colsum = nullptr;             // (B, rn_block128_nparts(B, T, num_cu), 128) partial sums of opool over pooled frames
    int B = 0, T = 0, Tout = 0, ntiles = 0;
    int per_wg = 0, nseg = 0;            // filled by the launcher: items per workgroup, workgroups that can share an utterance
    unsigned long long* dbg = nullptr;   // tools/rb_bench: per-workgroup phase cycle totals (only read in -DSVHIP_GEMM_DEBUG builds)
    int debug = 0;                       // tools/rb_bench ablations (debug builds): 1 no fragment reads, 2 no MFMA, 4 no conversion, 8 no pool
    int f16 = 0;                         // the 16-bit tensors (xin, W1, W2, opool) hold fp16, not bf16
};
int rn_block128_ntiles(int T);
int rn_block128_nparts(int B, int T, int num_cu);
bool rn_block128_supported(int cin, int cout, int T, bool downsample, bool has_shortcut, int Kp1, int Kp2);
hipError_t launch_rn_block128(const RnBlock128Params& p, int num_cu, hipStream_t stream);
// AFMS gate from partial column sums: s (B, C) = sigmoid(fc(sum(part) / Tn)); part (B, nparts, C), WT = fc weight TRANSPOSED [C][C] fp32
hipError_t launch_rn_afms_gate(const float* part, int nparts, int B, int C, int Tn, const float* WT, const float* bias, float* s,
                               hipStream_t stream);
hipError_t launch_rn_attn_pool(const float* logits, const void* x, int dt, int B, int T, int C, float* out, hipStream_t stream);

// synthetic waveforms from a counter-based RNG (synth.hip): out (B, L) fp32 = utterances [first_utt, first_utt + B) of the stream `seed`
hipError_t launch_synth_wave(float* out, uint64_t seed, int64_t first_utt, int B, int L, hipStream_t stream);

// ---------------------------------------------------------------------------------------------
// Scoring
// ---------------------------------------------------------------------------------------------
hipError_t launch_l2norm(float* E, int64_t N, int D, hipStream_t stream);
hipError_t launch_score_pairs(const float* E, int D, const int32_t* ia, const int32_t* ib, int64_t P, float* out, hipStream_t stream);
hipError_t launch_asnorm_pairs(const float* E, int D, const float* mu, const float* sigma, const int32_t* ia,
                               const int32_t* ib, int64_t P, float* out, hipStream_t stream);
// whole-trial forms over the aligned crops of two files, F (n_files, n_crops, D): mode 0 mean |cos|, 1 mean p-2 distance (+1e-6),
// 2 minus the mean pairwise distance over the (n, D, n) broadcast (src/utils.py:163-169, src/model.py:425-431)
hipError_t launch_trial_crops(int mode, float pexp, const float* F, int n_crops, int D, const int32_t* ia, const int32_t* ib, int64_t P, float* out,
                              hipStream_t stream);
hipError_t launch_mean_crops(const float* F, int64_t n_files, int n_crops, int D, float* out, hipStream_t stream);
// S (rows x K, row stride ld) fp32 cohort scores -> mean / population std of the `top` largest per row
hipError_t launch_topk_stats(const float* S, int64_t rows, int K, int ld, int top, float* mu, float* sigma, hipStream_t stream);

// Fused AS-norm cohort statistics (asnorm_fused.hip): scores stay in the MFMA accumulators; per embedding two candidate lists of
// ASNORM_CAND_PER_LANE scores above a moment-based threshold, then the exact top-`top` statistics of the candidates.
constexpr int ASNORM_CAND_PER_LANE = 256;
struct AsnormFusedParams {
    const float* E = nullptr;       // (N, D) embeddings of this launch
    int64_t N = 0;
    const float* cohort = nullptr;  // (K, D)
    int K = 0;
    const float* MB = nullptr;      // (D + 32, D): rows of M = C^T C / K, then the cohort mean, then zeros (launch_cohort_moments)
    float z = 0.0f;                 // threshold = row mean + z * row std (asnorm_tail_z)
    const float* zrow = nullptr;    // optional (N): a z of its own per embedding — the refit passes over the embeddings whose cohort scores the
                                    // normal quantile did not fit (round 6)
    float* cand = nullptr;          // (N, 2, ASNORM_CAND_PER_LANE) candidate scores
    int32_t* cnt = nullptr;         // (N, 2) scores above the threshold seen by each of the two lanes (may exceed the list size)
    const void* planes = nullptr;   // optional: [nplanes][D + 32 + K][D] 16-bit parts of [MB ; cohort] (launch_asnorm_planes): the split forms
    int nplanes = 2;                // 2: half hi | lo, three fp16 MFMAs per product block (default); 3: bf16 h | m | l, six bf16 MFMAs
    float* rowscale = nullptr;      // the 16-wide half-plane kernel with pscale: (N) factor that takes a row's candidates (stored in the scaled
                                    // domain) back to scores: asnorm_cand_stats multiplies mu and sigma by it
    const uint32_t* pscale = nullptr;   // the 16-wide half-plane kernel: max-|x| word of the cohort the planes were scaled by (launch_asnorm_planes)
    int nlists = 2;                 // candidate lists per embedding: 2 (cnt (N, 2), lists of ASNORM_CAND_PER_LANE), or 4 with nplanes = 2: the
                                    // 16-wide-MFMA kernel (cnt (N, 4), lists of ASNORM_CAND_PER_LANE / 2)
};
// dense score matrix on half planes / 16x16x32 fp16 MFMAs (asnorm_fused.hip): out (Na, ldo) = A (Na, D) . B (Nb, D)^T, D = 192 / 256
struct ScoreH3Params {
    const float* A = nullptr;
    int64_t Na = 0;
    const void* planes = nullptr;   // [2][Nb][D] half parts of B
    int Nb = 0;
    float* out = nullptr;
    int64_t ldo = 0;
    int per = 0;                    // blocks of 32 rows of B per workgroup (column slice)
    const uint32_t* pscale = nullptr;   // max-|x| word of B: the planes hold B * 2^k (asnorm_fused.hip, "operand scaling")
};
bool score_h3w_supported(int D, int64_t Na, int64_t Nb);
size_t score_h3w_planes_bytes(int D, int64_t Nb);
hipError_t launch_score_h3w(const float* A, int64_t Na, const float* B, int64_t Nb, int D, float* out, int64_t ldo, void* planes, int num_cu,
                            hipStream_t stream);
bool asnorm_fused6_supported(int D, int planes);
size_t asnorm_planes_bytes(int D, int K);
// pscale (two half planes only): a device word that receives the cohort's max |x|; the planes are then scaled by an exact power of two
hipError_t launch_asnorm_planes(const float* MB, const float* cohort, int K, int D, void* planes, hipStream_t stream, int nplanes = 2, uint32_t* pscale = nullptr);
bool asnorm_fused_supported(int D, int K, int top);
float asnorm_tail_z(int K, int top);
// `part`: cohort_moments_scratch_bytes(D) of scratch (slice partials, summed in a fixed order)
size_t cohort_moments_scratch_bytes(int D);
hipError_t launch_cohort_moments(const float* cohort, int K, int D, float* MB, float* part, hipStream_t stream);
hipError_t launch_asnorm_fused(const AsnormFusedParams& p, int D, hipStream_t stream);
// mu / sigma [row_base + r] for r < rows; embeddings that cannot be decided from their candidates: flagged[atomicAdd(nflag, 1)] = index
// (finfo, optional, parallel to `flagged`: the candidate count of a flagged embedding, bit 30 = one of its lists overflowed)
hipError_t launch_asnorm_cand_stats(const float* cand, const int32_t* cnt, int64_t rows, int top, float* mu, float* sigma, int64_t row_base,
                                    int32_t* flagged, int32_t* nflag, hipStream_t stream, int nlists = 2, const float* rowscale = nullptr, int32_t* finfo = nullptr);
hipError_t launch_gather_rows(const float* E, const int32_t* ids, int n, int D, float* out, hipStream_t stream);
// refit state of flagged embeddings (asnorm_fused.hip): structure of arrays [id | z | zp | lcp | zlo | zhi], stride = the row count
hipError_t launch_asnorm_refit_init(const int32_t* ids, const int32_t* info, int n, float z0, float target, float* soa, hipStream_t stream);
hipError_t launch_asnorm_refit_next(const float* in, int n_in, const int32_t* pos, const int32_t* info, int left, float target, float* out, hipStream_t stream);
hipError_t launch_scatter_stats(const float* m, const float* s, const int32_t* ids, int n, float* mu, float* sigma, hipStream_t stream);

// ---------------------------------------------------------------------------------------------
// Verification metrics (metrics.hip): one workspace of metrics_workspace_bytes(P) holds the sorted trial list
// ---------------------------------------------------------------------------------------------
size_t metrics_workspace_bytes(int64_t P);
hipError_t metrics_sort_scan(const float* scores, const int32_t* labels, int64_t P, bool nan_to_num, void* ws, size_t ws_bytes, hipStream_t st);
hipError_t metrics_roc_points(int64_t P, void* ws, size_t ws_bytes, float* thr, int64_t* fps, int64_t* tps, int32_t** n_runs_dev, hipStream_t st);
hipError_t metrics_error_rates(int64_t P, void* ws, size_t ws_bytes, double* fnrs, double* fprs, float* thresholds, hipStream_t st);
hipError_t metrics_min_dcf(int64_t P, void* ws, size_t ws_bytes, double p_target, double c_miss, double c_fa, double* dcf_dev, float* thr_dev,
                           hipStream_t st);

}  // namespace svhip
