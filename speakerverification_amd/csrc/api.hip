// api.hip — C ABI of libsvhip: handle, weight packing, forward orchestration (see include/svhip.h).
#include "../../include/svhip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "common.h"
#include "kernels.h"

using namespace svhip;

namespace {

thread_local std::string g_create_error;

struct HostTensor {
    std::vector<float> data;
    std::vector<int64_t> shape;
    int64_t numel() const { int64_t n = 1; for (auto s : shape) n *= s; return n; }
};

struct ConvLayer {            // one conv1d as a GEMM operand set (device pointers)
    int N = 0, K = 0, Kp = 0, Np = 0, taps = 1, dil = 1, cin = 0;
    void* W = nullptr;        // packed [Np][Kp] in the compute dtype
    void* Wsplit = nullptr;   // SVHIP_F32X3 handles: the same matrix as (hi bf16 << 16 | lo bf16) words, for gemm_pw's split path
    float cv_wscale = 1.0f;   // ... whose planes hold cv_wscale * W (an exact power of two; 1 unless max |w| lies outside [2^-8, 2^13))
    void* Wcv = nullptr;      // SVHIP_F32X3 handles, odd-tap convolutions with N % 256 == 0 (blocks.0): [N][cv_Kp] S32, k = tap * cv_cin + c with the
    int cv_cin = 0, cv_Kp = 0; // input channels zero-padded to cv_cin (a multiple of 32) and cv_Kp = taps * cv_cin rounded up to 64: gemm_pw3's CV form
    void* Ws32 = nullptr;     // SVHIP_F32X3 handles, pointwise layers with N % 256 == 0 and K % 64 == 0: the S32 split layout (per row, per
                              // 32 k: 32 hi bf16 | 32 lo bf16) of gemm_pw3's X3 form
    float* bias = nullptr;    // [N] or null
    float* scale = nullptr;   // folded BatchNorm (eval): y = x*scale + shift, or null
    float* shift = nullptr;
    double flops_per_row = 0;
};

struct LinearLayer {          // small-M fp32 linear (rowvec kernel)
    int N = 0, K = 0;
    float* W = nullptr;       // [N][K]
    float* bias = nullptr;
};

struct ProfEntry { std::string name; double ms = 0; int64_t launches = 0; double flops = 0; };
struct PendingEvent { hipEvent_t e0, e1; int entry; };

}  // namespace

struct svhip_handle {
    svhip_config cfg{};
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipStream_t cur = nullptr;                // stream the launch helpers enqueue on (main stream or a lane)
    hipStream_t lane_stream[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t lane_ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};       // [lane] done events, [4] = fork point
    int lanes = 1;                            // > 1: the forward runs as that many batch slices on as many streams
    // developer / test switches.  The SVHIP_* environment variables of the same names are read ONCE, in svhip_create (they are the
    // defaults of a new handle); afterwards only svhip_set_option changes them: no getenv on any forward or scoring call
    struct DevOpts {
        int layer_labels = 0;     // one profile row per GEMM shape
        int x3_keep_f32 = 0;      // F32X3: keep the fp32 copies of the block outputs beside the split layout
        int r2_big = 0;           // F32X3: Res2Net steps on the R2 form of the 256 x 256 kernel instead of r2_step
        int asp_v1 = 0;           // bf16: asp_fused_kernel instead of asp_bf16_kernel
        int rn_stop = -1;         // RawNet2: return after this many residual blocks (0: after the sinc front-end), unfused kernel sequence
        int rn_snap = -1;         // RawNet2: keep block n's pre-activation as stage "rn_snap"
        int rn_unfused = 0;       // RawNet2: the separate kernel sequence instead of rn_block128 / rn_tail / the folded shortcut
        int asnorm_slab = 0;      // AS-norm statistics on the slab path
        int asnorm_f32mfma = 0;   // AS-norm fused kernel on the exact fp32 MFMA instead of a split form
        int score_f32mfma = 0;    // dense score GEMMs (svhip_score_matrix, the slab path's cohort GEMM) on the exact fp32 MFMA instead of the split form
        int score_tiled = 0;      // dense score GEMMs on the tiled split kernel (gemm_pw) instead of the row-streaming one (score_h3w)
        int asnorm_w32 = 0;       // AS-norm two-half-plane kernel on the 32-wide MFMA (round 4's first form) instead of 16x16x32
        int asnorm_norefit = 0;   // AS-norm: embeddings the normal-quantile threshold does not fit go straight to the slab path (round 5's behaviour)
        int asnorm_2s = 0;        // AS-norm split forms: candidate statistics of chunk c on a second stream under the matrix kernel of chunk c + 1
        int asnorm_x6 = 0;        // AS-norm fused kernel on six bf16 MFMAs (three planes, round 3) instead of three fp16 MFMAs (two planes)
        int rn_sinc_full = 0;     // RawNet2 fp16 handles: the 251-tap sinc kernel (round 5) instead of the symmetric 126-tap form
        int fbank32 = 0;          // the 32-frame front-end kernel
        int ff_abl = 0;           // tools only: fused front-end phase ablations (FbankTables::ff_abl)
        int fbank_unfused = 0;    // bf16 handles: fbank -> prologue_stats -> prologue_apply (round 5) instead of the fused front-end
        int pw3_cus = -1;         // cap of the persistent GEMM grids (0: persistent kernels off)
        int pw4 = 0;              // plain pointwise bf16 layers with more tiles than workgroups on the four-wave kernel (gemm_pw4.hip)
        int pw3_tail_off = 0;     // persistent 16-bit GEMMs: the last partial round as whole tiles (round 4) instead of column halves
        int cv_off = 0;           // 16-bit handles: conv-gather GEMMs on the per-tile kernel instead of the persistent one
        int n128_off = 0;         // bf16: asp.tdnn on gemm_pw instead of gemm_n128
        int rn_pool_off = 0;      // F32X3 handles: conv2 of the long pooled blocks writes the un-pooled output, rn_maxpool3 pools it (tests)
        int rn_step_off = 0;      // F32X3 handles: the 128 -> 128 blocks' convolutions on the tiled in-register-split kernel (tests)
        int rn_sinc_f32 = 0;      // F32X3 handles: the sinc front-end on the exact fp32 MFMA (tests) instead of three fp16 MFMAs per product
        int rn_tail_big = 0;      // RawNet2 block tail: one workgroup per utterance at every batch size (tests)
        int r2_slices = -1;       // bf16 Res2Net chain: time slices per utterance (-1: by batch size, 0 / 1: whole utterances, n: forced)
    } opt;
    bool bf16 = false;                        // 16-bit storage handle: bf16, or fp16 when `f16` is set (the flag keeps its round-1 name)
    bool f16 = false;                         // SVHIP_F16: the 16-bit type is IEEE half (RawNet2)
    int dt = DT_F32;                          // DT_F32 / DT_BF16 / DT_F16: what the element-wise launchers are told
    bool x3 = false;                          // SVHIP_F32X3: fp32 handle whose conv GEMMs run as split-bf16 MFMA triples
    bool finalized = false;
    std::string err;
    std::map<std::string, HostTensor> host_w;
    std::vector<void*> allocs;               // everything hipMalloc'ed, freed in destroy

    int T = 0;                                // frames per utterance
    int esz = 4;                              // activation element size

    // front-end tables
    FbankTables fb;

    // ECAPA layers
    ConvLayer blocks0, mfa, asp_tdnn, asp_conv;
    ConvLayer tdnn1[3], tdnn2[3], res2[3][7];
    LinearLayer se1[3], se2[3], asp_ctx, fc;
    float* se2T[3] = {};                      // se_block.conv2 weight transposed to [128][C]
    void *se1_bf[3] = {}, *se2T_bf[3] = {};   // bf16 copies of both SE matrices (bf16 handles: half the L2 bytes per workgroup)
    float *aspbn_scale = nullptr, *aspbn_shift = nullptr;
    float *in_w = nullptr, *in_b = nullptr;   // instance norm affine

    // RawNet2 layers (front_proc='sinc', aggregate='asp'; RawNet2_custom.py:230-243)
    struct RnBlock {
        int cin = 0, cout = 0;
        bool downsample = false, has_shortcut = false;
        float *bn1_scale = nullptr, *bn1_shift = nullptr;
        ConvLayer conv1, conv2, shortcut;       // conv1 carries bn2 as its epilogue
        void* conv2sc_W = nullptr;              // bf16 handles: [Np][conv2.K + cin] = conv2 | 1 x 1 shortcut, one GEMM for both (gemm_pw2 A3)
        float* alpha = nullptr;
        LinearLayer afms_fc;
        float* afms_fcT = nullptr;              // fc weight transposed [cin][cout] (the gate kernel reads consecutive outputs per wave)
    };
    RnBlock rn_blocks[8];
    float *rn_gamma = nullptr, *rn_beta = nullptr, *rn_fbn_scale = nullptr, *rn_fbn_shift = nullptr;
    void* rn_filt = nullptr;
    void* rn_filt_sym = nullptr;              // fp16 handles: [128][128] slot-major table of the symmetric sinc form (round 6)
    void* rn_filt_x3 = nullptr;               // F32X3 handles: [2][128][256] half hi | lo parts of the sinc filters
    float *rn_agg_scale = nullptr, *rn_agg_shift = nullptr;
    ConvLayer rn_att0, rn_att3;
    LinearLayer rn_fc;
    void* rn_buf[6] = {};                 // activation ping-pong buffers
    float* rn_scratch = nullptr;
    void* rn_xn = nullptr;
    int rn_Lp = 0;
    float *rn_stats = nullptr, *rn_mean = nullptr, *rn_s = nullptr, *rn_logits = nullptr, *rn_pooled = nullptr;
    float* rn_part = nullptr;             // fused 128-channel blocks: per-tile column sums (B, ntiles, 128)
    int num_cu = 256;
    int rn_T1 = 0;
    const void* rn_dbg_x = nullptr; int rn_dbg_T = 0, rn_dbg_C = 0;   // SVHIP_RN_STOP developer hook (tests)
    void* rn_snap = nullptr; size_t rn_snap_cap = 0; int rn_snap_T = 0, rn_snap_C = 0;      // SVHIP_RN_SNAP=2: copy of block 2's pre-activation (stage "rn_snap")

    // workspace (device)
    float* d_wav = nullptr;       // (Bmax, L)
    float* d_feat = nullptr;      // (Bmax, n_mels, T) mel power
    float* d_pstats = nullptr;    // (Bmax*n_mels*2)
    float* d_xscale = nullptr;    // F32X3: [0] = s, [1] = 1 / s of the network input (launch_in_scale), then 256 partial max words
    float* d_logmel = nullptr;    // fused front-end (bf16 handles): (Bmax, T, n_mels) log-mel rows before the mean is taken off
    float* d_fpart = nullptr;     //   and their per-tile column sums (Bmax, ceil(T / 64), n_mels)
    bool xin_ready = false;       // the fused front-end has written X_in: ecapa_forward_part skips its prologue
    bool feat_is_stale = false;   // ... and d_feat does not hold this forward's mel power (svhip_get_stage "mel")
    float* d_zero = nullptr;      // 256 zero bytes (DMA source for padded conv chunks)
    float *d_ones = nullptr, *d_zeros = nullptr;      // 4096 ones / zeros: stand-ins for absent per-channel vectors (GemmParams::ones / zeros)
    size_t rn_buf_bytes = 0;      // RawNet2: payload bytes of each activation buffer; a 256-byte zero tail follows (the zero page of the
                                  // persistent conv-gather kernel must sit behind its A operand, within 4 GiB)
    void* s32_buf = nullptr;      // SVHIP_F32X3: the A operand of the current big GEMM in the S32 split layout (M x 3C x 4 bytes)
    void *side_a = nullptr, *side_b = nullptr;      // pending S32 side outputs of the next conv_gemm (GemmParams::side_*), consumed by it
    int side_lda = 0, side_ldb = 0, side_c = 0;
    bool side_done = false;       // ... and whether that GEMM wrote them
    bool x0_is_s32 = false;       // SVHIP_F32X3: the last forward wrote blocks.0's output (X0) in the split layout
    bool cat_f32_stale = false;   // SVHIP_F32X3: the last forward left the block outputs only in cat_s32 (svhip_get_stage converts on demand)
    void* cat_s32 = nullptr;      // SVHIP_F32X3: the SE-Res2Net block outputs (the CAT buffer) in the S32 layout, written by se_apply
    void* h2_s32 = nullptr;       // SVHIP_F32X3: the Res2Net chain output (H2's twin, S32 only) and the two step-input buffers (M x C/8)
    void* u_s32[2] = {};
    float* d_colsum = nullptr;    // pw2 column-sum partials, per lane: [sum | sumsq] x (tiles*4) x 3C floats
    int64_t colsum_region = 0;    // floats per (lane, kind) region
    bool last_colsum_done = false;
    int last_colsum_groups = 8;   // row groups per tile in the partials the last GEMM wrote (8: pw2, 2: pw3)
    void* X_in = nullptr;         // (M, n_mels)
    void* X0 = nullptr;           // (M, C)
    void *H1 = nullptr, *H2 = nullptr, *H3 = nullptr;   // (M, C)
    void* CAT = nullptr;          // (M, 3C)
    void* MFA = nullptr;          // (M, 3C)
    void* ATT = nullptr;          // (M, 128)
    float* LOGITS = nullptr;      // (M, 3C) fp32
    float *d_mean = nullptr, *d_s1 = nullptr, *d_s2 = nullptr, *d_gstats = nullptr, *d_ctx = nullptr;
    float* d_lin_part = nullptr;              // K-slice partials of the small-M linear layers (fc, asp_ctx) at full batches
    size_t lin_part_per_utt = 0;
    float *d_pool_raw = nullptr, *d_pool_bn = nullptr, *d_emb = nullptr;
    int lastB = 0;
    // numeric status of the forwards since the last reset: d_status[0] = SVHIP_STATUS_* bits, [1] = non-finite embedding values,
    // [2] = input values beyond the split planes' range; host_flag (pinned, mapped) is set by the same kernels, so that a synchronous
    // call learns of a problem without a copy
    uint32_t* d_status = nullptr;
    uint32_t* host_flag = nullptr;
    uint32_t* host_flag_dev = nullptr;

    // profiling: event pairs are recorded around every launch without blocking the host and
    // resolved (hipEventElapsedTime) when results are read
    bool prof = false;
    std::string prof_filter;                  // non-empty: only launches with exactly this label are bracketed by events
    std::vector<hipEvent_t> ev_free;
    std::vector<PendingEvent> ev_pending;
    std::vector<ProfEntry> prof_entries;
    double flops_per_utt = 0;
    void* comm = nullptr;                     // RCCL communicator state, owned by comm.hip
    // svhip_crop_pcm16 staging (host-pointer calls): one grow-only device PCM buffer (copies and kernels are ordered on the
    // handle's stream) and a ring of pinned host / device metadata slots, each guarded by an event, so that SVHIP_ASYNC calls
    // can return before the copy has run
    void* crop_pcm = nullptr; size_t crop_pcm_cap = 0;
    // scoring / metrics scratch: handle-owned slots, grown on demand (no hipMalloc / hipFree per call once warm)
    enum { SCR_IN0 = 0, SCR_IN1, SCR_IN2, SCR_IN3, SCR_IN4, SCR_OUT0, SCR_OUT1, SCR_OUT2, SCR_SLAB, SCR_SPLIT, SCR_CAND, SCR_CNT, SCR_MB,
           SCR_FLAG, SCR_GATHER, SCR_WS, SCR_COUNT };
    void* scr[SCR_COUNT] = {};
    size_t scr_cap[SCR_COUNT] = {};
    hipStream_t aux_stream = nullptr;         // second stream of the scoring entry points (candidate statistics under the next MFMA launch)
    hipEvent_t aux_ev[4] = {};
    int64_t last_asnorm_refit = 0;            // embeddings of the last call that the refit passes of the fused kernel decided (round 6)
    int last_asnorm_refit_passes = 0;
    int last_asnorm_flagged = -1;             // embeddings the fused AS-norm kernel handed to the slab path in the last call (-1: slab path)
    struct CropSlot { char* host = nullptr; char* dev = nullptr; size_t cap = 0; hipEvent_t done = nullptr; bool busy = false; };
    CropSlot crop_slot[4];
    int crop_next = 0;
};

namespace {

#define SV_FAIL(h, code, ...)                                   \
    do {                                                        \
        char _b[512];                                           \
        snprintf(_b, sizeof(_b), __VA_ARGS__);                  \
        (h)->err = _b;                                          \
        return (code);                                          \
    } while (0)

#define SV_HIP(h, expr)                                                                            \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) SV_FAIL(h, SVHIP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

template <typename T>
int dev_alloc(svhip_handle* h, T** p, size_t count) {
    void* q = nullptr;
    size_t bytes = count * sizeof(T);
    if (bytes == 0) bytes = 16;
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) SV_FAIL(h, SVHIP_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    h->allocs.push_back(q);
    *p = reinterpret_cast<T*>(q);
    return SVHIP_OK;
}

template <typename T>
int dev_upload(svhip_handle* h, T** p, const std::vector<T>& v) {
    int rc = dev_alloc(h, p, v.size());
    if (rc) return rc;
    if (!v.empty()) SV_HIP(h, hipMemcpy(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return SVHIP_OK;
}

inline uint16_t f32_to_f16_rne(float f) {      // IEEE half, round to nearest even (the host compiler's _Float16 conversion)
    const _Float16 hv = static_cast<_Float16>(f);
    uint16_t u;
    memcpy(&u, &hv, 2);
    return u;
}

inline uint16_t f32_to_bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// host twin of common.h's x3_hi / x3_lo: a weight as (hi << 16) | lo in the planes' type (IEEE half; bf16 under -DSVHIP_X3_BF16)
inline uint32_t x3_split_word(float v) {
#ifdef SVHIP_X3_BF16
    const uint16_t hi = f32_to_bf16_rne(v);
    uint32_t hu = (uint32_t)hi << 16;
    float hf; memcpy(&hf, &hu, 4);
    return hu | f32_to_bf16_rne(v - hf);
#else
    const _Float16 h = static_cast<_Float16>(v);                 // (plain conversions: overflow -> inf, NaN stays NaN — common.h, RANGE)
    const _Float16 l = static_cast<_Float16>(v - static_cast<float>(h));
    uint16_t hb, lb;
    memcpy(&hb, &h, 2); memcpy(&lb, &l, 2);
    return ((uint32_t)hb << 16) | lb;
#endif
}

// a weight in the handle's 16-bit storage type
inline uint16_t to_h16(const svhip_handle* h, float f) { return h->f16 ? f32_to_f16_rne(f) : f32_to_bf16_rne(f); }

// ---- profiling-aware launch wrapper --------------------------------------------------------------
hipEvent_t prof_event(svhip_handle* h) {
    if (!h->ev_free.empty()) { hipEvent_t e = h->ev_free.back(); h->ev_free.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

void prof_collect(svhip_handle* h) {
    if (h->ev_pending.empty()) return;
    (void)hipStreamSynchronize(h->stream);
    for (int i = 0; i < 4; ++i) if (h->lane_stream[i]) (void)hipStreamSynchronize(h->lane_stream[i]);
    for (auto& pe : h->ev_pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, pe.e0, pe.e1) == hipSuccess) h->prof_entries[pe.entry].ms += ms;
        h->ev_free.push_back(pe.e0);
        h->ev_free.push_back(pe.e1);
    }
    h->ev_pending.clear();
}

template <typename F>
int run(svhip_handle* h, const char* label, double flops, F&& launch) {
    PendingEvent pe{nullptr, nullptr, -1};
    const bool prof = h->prof && (h->prof_filter.empty() || h->prof_filter == label);
    if (prof) {
        for (size_t i = 0; i < h->prof_entries.size(); ++i)
            if (h->prof_entries[i].name == label) { pe.entry = (int)i; break; }
        if (pe.entry < 0) { h->prof_entries.push_back(ProfEntry{label}); pe.entry = (int)h->prof_entries.size() - 1; }
        pe.e0 = prof_event(h);
        pe.e1 = prof_event(h);
        (void)hipEventRecord(pe.e0, h->cur);
    }
    hipError_t e = launch();
    if (e != hipSuccess) SV_FAIL(h, SVHIP_ERR_HIP, "launch %s failed: %s", label, hipGetErrorString(e));
    if (prof) {
        (void)hipEventRecord(pe.e1, h->cur);
        h->prof_entries[pe.entry].launches += 1;
        h->prof_entries[pe.entry].flops += flops;
        h->ev_pending.push_back(pe);
        if (h->ev_pending.size() >= 8192) prof_collect(h);
    }
    return SVHIP_OK;
}

// ---- front-end tables (oracle/fbank.py restates the same constants) --------------------------------
double hz_to_mel(double f) {
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
    return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : f / f_sp;
}
double mel_to_hz(double m) {
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
    return m >= min_log_mel ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : f_sp * m;
}

int build_fbank_tables(svhip_handle* h) {
    const svhip_config& c = h->cfg;
    FbankTables& fb = h->fb;
    fb.n_fft = c.n_fft; fb.win_length = c.win_length; fb.hop = c.hop_length; fb.n_mels = c.n_mels;
    fb.n_bins = c.n_fft / 2 + 1;
    fb.lpad = (c.n_fft - c.win_length) / 2;
    fb.n_pairs = (fb.n_bins + 31) / 32;
    fb.n_q = c.win_length / 8;
    fb.preemph = c.preemph;
    if (c.win_length % 8 != 0 || c.hop_length % 4 != 0 || fb.n_pairs > 9 || c.win_length > c.n_fft)
        SV_FAIL(h, SVHIP_ERR_UNSUPPORTED, "fbank geometry n_fft=%d win=%d hop=%d not supported", c.n_fft, c.win_length, c.hop_length);
    const double PI = 3.14159265358979323846;
    // periodic Hamming (scipy get_window('hamming', win, fftbins=True)), cast to float32
    std::vector<float> win(c.win_length);
    for (int k = 0; k < c.win_length; ++k) win[k] = (float)(0.54 - 0.46 * std::cos(2.0 * PI * k / c.win_length));
    // basis[q][pair][part][lane] float4: tap = 8q + 4h + j, bin = 32*pair + r (lane = 32h + r); part 0 = cos, 1 = sin
    std::vector<float> basis((size_t)fb.n_q * fb.n_pairs * 2 * 64 * 4, 0.0f);
    for (int q = 0; q < fb.n_q; ++q)
        for (int pr = 0; pr < fb.n_pairs; ++pr)
            for (int part = 0; part < 2; ++part)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 4; ++j) {
                        const int r = lane & 31, hh = lane >> 5;
                        const int tap = 8 * q + 4 * hh + j, bin = 32 * pr + r;
                        float v = 0.0f;
                        if (bin < fb.n_bins) {
                            const double ang = 2.0 * PI * (double)bin * (double)(fb.lpad + tap) / (double)c.n_fft;
                            const float tr = (float)(part == 0 ? std::cos(ang) : std::sin(ang));
                            v = tr * win[tap];                       // float32 product, as nnAudio's kernel * window mask
                        }
                        basis[((((size_t)q * fb.n_pairs + pr) * 2 + part) * 64 + lane) * 4 + j] = v;
                    }
    // bf16x3 tables: the same windowed taps split into bf16 hi + lo, k-steps of 16 (zero padded)
    fb.n_k16 = (c.win_length + 15) / 16;
    fb.split_bf16 = (h->bf16 && c.hop_length % 8 == 0) ? 1 : 0;
    fb.split6 = (h->x3 && c.hop_length % 8 == 0) ? 1 : 0;          // F32X3 handles: the exact three-way split, six products
    if (fb.split_bf16 || fb.split6) {
        std::vector<uint16_t> bhi((size_t)fb.n_k16 * fb.n_pairs * 2 * 64 * 8, 0), blo(bhi.size(), 0), bl3(bhi.size(), 0);
        for (int kk = 0; kk < fb.n_k16; ++kk)
            for (int pr = 0; pr < fb.n_pairs; ++pr)
                for (int part = 0; part < 2; ++part)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int r = lane & 31, hh = lane >> 5;
                            const int tap = 16 * kk + 8 * hh + j, bin = 32 * pr + r;
                            float v = 0.0f;
                            if (bin < fb.n_bins && tap < c.win_length) {
                                const double ang = 2.0 * PI * (double)bin * (double)(fb.lpad + tap) / (double)c.n_fft;
                                v = (float)(part == 0 ? std::cos(ang) : std::sin(ang)) * win[tap];
                            }
                            const uint16_t hi = f32_to_bf16_rne(v);
                            uint32_t hu = (uint32_t)hi << 16;
                            float hf; memcpy(&hf, &hu, 4);
                            const size_t idx = ((((size_t)kk * fb.n_pairs + pr) * 2 + part) * 64 + lane) * 8 + j;
                            bhi[idx] = hi;
                            blo[idx] = f32_to_bf16_rne(v - hf);
                            uint32_t mu = (uint32_t)blo[idx] << 16;
                            float mf; memcpy(&mf, &mu, 4);
                            bl3[idx] = f32_to_bf16_rne((v - hf) - mf);
                        }
        uint16_t *dh, *dl;
        int rc2;
        if ((rc2 = dev_upload(h, &dh, bhi))) return rc2;
        if ((rc2 = dev_upload(h, &dl, blo))) return rc2;
        fb.basis_hi = dh; fb.basis_lo = dl;
        if (fb.split6) {
            uint16_t* d3;
            if ((rc2 = dev_upload(h, &d3, bl3))) return rc2;
            fb.basis_l3 = d3;
        }
    }
    // the fused front-end of bf16 handles (fbank.hip, round 6): the window is symmetric about tap win / 2, so Re X_k / Im X_k are products of
    // K = win / 2 + 1 taps with w_m cos(2 pi k m / n_fft) / w_m sin(2 pi k m / n_fft), m = 0 .. win / 2 (slot win / 2 carries the unpaired tap 0)
    if (h->bf16 && c.n_fft == 512 && c.win_length == 200 && c.hop_length == 80) {
        const int half = c.win_length / 2, nks = 7, npr = 8;
        std::vector<uint16_t> shi((size_t)nks * npr * 2 * 64 * 8, 0), slo(shi.size(), 0);
        for (int kk = 0; kk < nks; ++kk)
            for (int pr = 0; pr < npr; ++pr)
                for (int part = 0; part < 2; ++part)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int r = lane & 31, hh = lane >> 5;
                            const int m = 16 * kk + 8 * hh + j, bin = 32 * pr + r;
                            float v = 0.0f;
                            if (m <= half && bin < fb.n_bins) {
                                const float w = m < half ? win[half + m] : win[0];
                                const double ang = 2.0 * PI * (double)bin * (double)m / (double)c.n_fft;
                                v = (float)(part == 0 ? std::cos(ang) : std::sin(ang)) * w;
                            }
                            const uint16_t hi = f32_to_bf16_rne(v);
                            uint32_t hu = (uint32_t)hi << 16;
                            float hf; memcpy(&hf, &hu, 4);
                            const size_t idx = ((((size_t)kk * npr + pr) * 2 + part) * 64 + lane) * 8 + j;
                            shi[idx] = hi;
                            slo[idx] = f32_to_bf16_rne(v - hf);
                        }
        uint16_t *dh, *dl;
        int rc2;
        if ((rc2 = dev_upload(h, &dh, shi))) return rc2;
        if ((rc2 = dev_upload(h, &dl, slo))) return rc2;
        fb.sym_hi = dh; fb.sym_lo = dl;
    }
    // Slaney mel bank (librosa 0.7 filters.mel(htk=False, norm=1)) in double, stored float32, sparse rows
    const double sr = c.fb_sr;
    const double fmax = c.fmax > 0 ? c.fmax : sr / 2;
    const int nm = c.n_mels, nb = fb.n_bins;
    std::vector<double> mel_f(nm + 2), fftf(nb);
    for (int i = 0; i < nb; ++i) fftf[i] = (sr / 2) * i / (double)(nb - 1);
    const double m0 = hz_to_mel(c.fmin), m1 = hz_to_mel(fmax);
    for (int i = 0; i < nm + 2; ++i) mel_f[i] = mel_to_hz(m0 + (m1 - m0) * i / (double)(nm + 1));
    std::vector<float> mw;
    std::vector<int> mstart(nm), mlen(nm), moff(nm);
    for (int i = 0; i < nm; ++i) {
        const double fd0 = mel_f[i + 1] - mel_f[i], fd1 = mel_f[i + 2] - mel_f[i + 1];
        const double enorm = 2.0 / (mel_f[i + 2] - mel_f[i]);
        int first = -1, last = -1;
        std::vector<float> row(nb);
        for (int k = 0; k < nb; ++k) {
            const double lower = -(mel_f[i] - fftf[k]) / fd0, upper = (mel_f[i + 2] - fftf[k]) / fd1;
            const float w32 = (float)std::fmax(0.0, std::fmin(lower, upper));
            row[k] = (float)((double)w32 * enorm);
            if (row[k] != 0.0f) { if (first < 0) first = k; last = k; }
        }
        if (first < 0) { first = 0; last = -1; }
        mstart[i] = first; mlen[i] = last - first + 1; moff[i] = (int)mw.size();
        for (int k = first; k <= last; ++k) mw.push_back(row[k]);
    }
    if (mw.empty()) mw.push_back(0.0f);
    float* d_basis; float* d_mw; int *d_ms, *d_ml, *d_mo;
    int rc;
    if ((rc = dev_upload(h, &d_basis, basis))) return rc;
    if ((rc = dev_upload(h, &d_mw, mw))) return rc;
    if ((rc = dev_upload(h, &d_ms, mstart))) return rc;
    if ((rc = dev_upload(h, &d_ml, mlen))) return rc;
    if ((rc = dev_upload(h, &d_mo, moff))) return rc;
    fb.n_melw = (int)mw.size();
    fb.mel_max_bin = 0;
    for (int i = 0; i < nm; ++i) fb.mel_max_bin = std::max(fb.mel_max_bin, mstart[i] + mlen[i] - 1);
    fb.basis = d_basis; fb.mel_w = d_mw; fb.mel_start = d_ms; fb.mel_len = d_ml; fb.mel_off = d_mo;
    return SVHIP_OK;
}

// ---- expected weight names / shapes ----------------------------------------------------------------
const int ECAPA_K[5] = {5, 3, 3, 3, 1};
const int ECAPA_D[5] = {1, 2, 3, 4, 1};

void ecapa_spec(const svhip_config& c, std::map<std::string, std::vector<int64_t>>& spec) {
    const int64_t C = c.channels, C3 = 3 * C, nm = c.n_mels;
    auto bn = [&](const std::string& p, int64_t n) {
        spec[p + ".weight"] = {n}; spec[p + ".bias"] = {n}; spec[p + ".running_mean"] = {n};
        spec[p + ".running_var"] = {n}; spec[p + ".num_batches_tracked"] = {};
    };
    auto tdnn = [&](const std::string& p, int64_t cin, int64_t cout, int64_t k) {
        spec[p + ".conv.conv.weight"] = {cout, cin, k}; spec[p + ".conv.conv.bias"] = {cout};
        bn(p + ".norm.norm", cout);
    };
    if (c.input_norm) { spec["instance_norm.weight"] = {nm}; spec["instance_norm.bias"] = {nm}; }
    tdnn("blocks.0", nm, C, ECAPA_K[0]);
    for (int i = 1; i <= 3; ++i) {
        const std::string p = "blocks." + std::to_string(i);
        tdnn(p + ".tdnn1", C, C, 1);
        for (int j = 0; j < 7; ++j) tdnn(p + ".res2net_block.blocks." + std::to_string(j), C / 8, C / 8, ECAPA_K[i]);
        tdnn(p + ".tdnn2", C, C, 1);
        spec[p + ".se_block.conv1.conv.weight"] = {128, C, 1}; spec[p + ".se_block.conv1.conv.bias"] = {128};
        spec[p + ".se_block.conv2.conv.weight"] = {C, 128, 1}; spec[p + ".se_block.conv2.conv.bias"] = {C};
    }
    tdnn("mfa", C3, C3, 1);
    tdnn("asp.tdnn", 3 * C3, 128, 1);
    spec["asp.conv.conv.weight"] = {C3, 128, 1}; spec["asp.conv.conv.bias"] = {C3};
    bn("asp_bn.norm", 2 * C3);
    spec["fc.conv.weight"] = {(int64_t)c.embed_dim, 2 * C3, 1}; spec["fc.conv.bias"] = {(int64_t)c.embed_dim};
}

const int RN_LAYERS[6] = {1, 1, 1, 2, 1, 2};                   // RawNet2_custom.py:231
const int RN_FILTERS[6] = {128, 128, 256, 256, 512, 512};      // RawNet2_custom.py:232

void rawnet2_spec(const svhip_config& c, std::map<std::string, std::vector<int64_t>>& spec) {
    auto bn = [&](const std::string& p, int64_t n) {
        spec[p + ".weight"] = {n}; spec[p + ".bias"] = {n}; spec[p + ".running_mean"] = {n};
        spec[p + ".running_var"] = {n}; spec[p + ".num_batches_tracked"] = {};
    };
    spec["ln.gamma"] = {(int64_t)c.samples}; spec["ln.beta"] = {(int64_t)c.samples};
    spec["first_conv.low_hz_"] = {128, 1}; spec["first_conv.band_hz_"] = {128, 1};
    bn("first_bn", 128);
    int64_t inpl = 128;
    for (int li = 0; li < 6; ++li)
        for (int b = 0; b < RN_LAYERS[li]; ++b) {
            const std::string p = "layer" + std::to_string(li + 1) + "." + std::to_string(b);
            const int64_t planes = RN_FILTERS[li];
            bn(p + ".bn1", inpl);
            spec[p + ".conv1.weight"] = {planes, inpl, 3};
            bn(p + ".bn2", planes);
            spec[p + ".conv2.weight"] = {planes, planes, 3};
            spec[p + ".afms.alpha"] = {planes, 1};
            spec[p + ".afms.fc.weight"] = {planes, planes}; spec[p + ".afms.fc.bias"] = {planes};
            if (inpl != planes) spec[p + ".shortcut.0.weight"] = {planes, inpl, 1};
            inpl = planes;
        }
    bn("bn_before_agg", 512);
    spec["attention.0.weight"] = {128, 512, 1}; spec["attention.0.bias"] = {128};
    bn("attention.2", 128);
    spec["attention.3.weight"] = {512, 128, 1}; spec["attention.3.bias"] = {512};
    spec["fc.weight"] = {(int64_t)c.embed_dim, 1024}; spec["fc.bias"] = {(int64_t)c.embed_dim};
}

void model_spec(const svhip_config& c, std::map<std::string, std::vector<int64_t>>& spec) {
    if (c.model == SVHIP_MODEL_ECAPA) ecapa_spec(c, spec);
    else if (c.model == SVHIP_MODEL_RAWNET2) rawnet2_spec(c, spec);
}

const HostTensor* getw(svhip_handle* h, const std::string& name) {
    auto it = h->host_w.find(name);
    return it == h->host_w.end() ? nullptr : &it->second;
}

// fold BatchNorm1d(eval, eps=1e-5) into scale / shift (double arithmetic on the host)
int make_bn(svhip_handle* h, const std::string& p, int n, float** scale, float** shift) {
    const HostTensor *w = getw(h, p + ".weight"), *b = getw(h, p + ".bias"), *rm = getw(h, p + ".running_mean"),
                     *rv = getw(h, p + ".running_var");
    if (!w || !b || !rm || !rv) SV_FAIL(h, SVHIP_ERR_MISSING, "missing BatchNorm tensors for %s", p.c_str());
    std::vector<float> sc(n), sh(n);
    for (int i = 0; i < n; ++i) {
        const double s = (double)w->data[i] / std::sqrt((double)rv->data[i] + 1e-5);
        sc[i] = (float)s;
        sh[i] = (float)((double)b->data[i] - (double)rm->data[i] * s);
    }
    int rc;
    if ((rc = dev_upload(h, scale, sc))) return rc;
    return dev_upload(h, shift, sh);
}

// pack conv weight (N, cin, taps) columns [c_lo, c_hi) -> [Np][Kp], k = tap*cin' + c
int make_conv(svhip_handle* h, ConvLayer& L, const std::string& wname, const std::string& bname, const std::string& bnname,
              int dil, int c_lo = 0, int c_hi = -1) {
    const HostTensor* w = getw(h, wname);
    if (!w) SV_FAIL(h, SVHIP_ERR_MISSING, "missing tensor %s", wname.c_str());
    const int N = (int)w->shape[0], cin_full = (int)w->shape[1], taps = (int)w->shape[2];
    if (c_hi < 0) c_hi = cin_full;
    const int cin = c_hi - c_lo;
    const int bk = gemm_bk(h->bf16);
    L.N = N; L.taps = taps; L.dil = dil; L.cin = cin; L.K = taps * cin;
    L.Kp = round_up(L.K, bk); L.Np = round_up(N, GEMM_BN);
    L.flops_per_row = 2.0 * N * L.K;
    std::vector<float> packed((size_t)L.Np * L.Kp, 0.0f);
    for (int n = 0; n < N; ++n)
        for (int t = 0; t < taps; ++t)
            for (int c = 0; c < cin; ++c)
                packed[(size_t)n * L.Kp + t * cin + c] = w->data[((size_t)n * cin_full + (c_lo + c)) * taps + t];
    int rc;
    if (h->bf16) {
        std::vector<uint16_t> pb(packed.size());
        for (size_t i = 0; i < packed.size(); ++i) pb[i] = to_h16(h, packed[i]);
        uint16_t* d;
        if ((rc = dev_upload(h, &d, pb))) return rc;
        L.W = d;
    } else {
        float* d;
        if ((rc = dev_upload(h, &d, packed))) return rc;
        L.W = d;
        if (h->x3) {
            std::vector<uint32_t> ws(packed.size());
            for (size_t i = 0; i < packed.size(); ++i) ws[i] = x3_split_word(packed[i]);      // (hi plane << 16) | lo plane, x3_t of common.h
            uint32_t* dsplit;
            if ((rc = dev_upload(h, &dsplit, ws))) return rc;
            L.Wsplit = dsplit;
            // pointwise GELU layers (gemm_pw3's X3 form) and the Res2Net convolutions (its R2 form: N == cin, k = 3)
            if ((taps == 1 && N % 256 == 0 && L.K == L.Kp && L.K % 64 == 0 && L.K >= 128) ||
                (taps == 3 && N == cin && (cin == 64 || cin == 128) && L.K == L.Kp) ||
                // RawNet2's convolutions and projection shortcuts (r2_step.hip, modes 1 / 2)
                (h->cfg.model == SVHIP_MODEL_RAWNET2 && (taps == 1 || taps == 3) && N % 128 == 0 && cin % 32 == 0 && L.K == L.Kp && L.K == taps * cin)) {
                std::vector<uint16_t> s32((size_t)N * L.K * 2);
                for (int n = 0; n < N; ++n)
                    for (int k = 0; k < L.K; ++k) {
                        const uint32_t wv = ws[(size_t)n * L.Kp + k];
                        const size_t o = (size_t)n * L.K * 2 + (size_t)(k >> 5) * 64 + (k & 31);
                        s32[o] = (uint16_t)(wv >> 16);
                        s32[o + 32] = (uint16_t)(wv & 0xffffu);
                    }
                uint16_t* d32;
                if ((rc = dev_upload(h, &d32, s32))) return rc;
                L.Ws32 = d32;
            }
            if (taps >= 3 && taps <= 7 && (taps & 1) && N % 256 == 0 && N != cin) {      // the conv-gather X3 form (gemm_pw3cv)
                const int ccv = round_up(cin, 32), kcv = round_up(taps * ccv, 64);
                // the first convolution of the network meets features of whatever magnitude the checkpoint was trained on: weights fitted to
                // int16-scaled mel power are ~1e-10 — below the half planes' resolution.  Outside the ordinary range the planes hold sw * W,
                // sw an exact power of two (max |w| -> [64, 128)); the kernel multiplies back together with the input's scale (GemmParams::in_scale)
                float wmax = 0.0f;
                for (int n = 0; n < N; ++n)
                    for (int k = 0; k < L.K; ++k) { const float a = std::fabs(packed[(size_t)n * L.Kp + k]); if (std::isfinite(a) && a > wmax) wmax = a; }
                float sw = 1.0f;
                if (wmax > 0.0f && !(wmax >= 0x1p-8f && wmax < 0x1p13f)) { int e2; (void)std::frexp(wmax, &e2); sw = std::ldexp(1.0f, 7 - e2); }
                L.cv_wscale = sw;
                std::vector<uint16_t> s32((size_t)N * kcv * 2, 0);
                for (int n = 0; n < N; ++n)
                    for (int t = 0; t < taps; ++t)
                        for (int c = 0; c < cin; ++c) {
                            const uint32_t wv = sw == 1.0f ? ws[(size_t)n * L.Kp + t * cin + c] : x3_split_word(packed[(size_t)n * L.Kp + t * cin + c] * sw);
                            const int k = t * ccv + c;
                            const size_t o = (size_t)n * kcv * 2 + (size_t)(k >> 5) * 64 + (k & 31);
                            s32[o] = (uint16_t)(wv >> 16);
                            s32[o + 32] = (uint16_t)(wv & 0xffffu);
                        }
                uint16_t* dcv;
                if ((rc = dev_upload(h, &dcv, s32))) return rc;
                L.Wcv = dcv; L.cv_cin = ccv; L.cv_Kp = kcv;
            }
        }
    }
    if (!bname.empty()) {
        const HostTensor* b = getw(h, bname);
        if (!b) SV_FAIL(h, SVHIP_ERR_MISSING, "missing tensor %s", bname.c_str());
        if ((rc = dev_upload(h, &L.bias, b->data))) return rc;
    }
    if (!bnname.empty()) return make_bn(h, bnname, N, &L.scale, &L.shift);
    return SVHIP_OK;
}

int make_tdnn(svhip_handle* h, ConvLayer& L, const std::string& p, int dil) {
    return make_conv(h, L, p + ".conv.conv.weight", p + ".conv.conv.bias", p + ".norm.norm", dil);
}

// fp32 linear from a (N, K, 1) or (N, K) tensor, optional column range
int make_linear(svhip_handle* h, LinearLayer& L, const std::string& wname, const std::string& bname, int c_lo = 0, int c_hi = -1) {
    const HostTensor* w = getw(h, wname);
    if (!w) SV_FAIL(h, SVHIP_ERR_MISSING, "missing tensor %s", wname.c_str());
    const int N = (int)w->shape[0], Kfull = (int)w->shape[1];
    if (c_hi < 0) c_hi = Kfull;
    L.N = N; L.K = c_hi - c_lo;
    std::vector<float> m((size_t)N * L.K);
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < L.K; ++k) m[(size_t)n * L.K + k] = w->data[(size_t)n * Kfull + c_lo + k];
    int rc;
    if ((rc = dev_upload(h, &L.W, m))) return rc;
    if (!bname.empty()) {
        const HostTensor* b = getw(h, bname);
        if (!b) SV_FAIL(h, SVHIP_ERR_MISSING, "missing tensor %s", bname.c_str());
        if ((rc = dev_upload(h, &L.bias, b->data))) return rc;
    }
    return SVHIP_OK;
}

int finalize_ecapa(svhip_handle* h) {
    const int C = h->cfg.channels, C3 = 3 * C;
    int rc;
    if ((rc = make_tdnn(h, h->blocks0, "blocks.0", ECAPA_D[0]))) return rc;
    for (int i = 1; i <= 3; ++i) {
        const std::string p = "blocks." + std::to_string(i);
        if ((rc = make_tdnn(h, h->tdnn1[i - 1], p + ".tdnn1", 1))) return rc;
        for (int j = 0; j < 7; ++j)
            if ((rc = make_tdnn(h, h->res2[i - 1][j], p + ".res2net_block.blocks." + std::to_string(j), ECAPA_D[i]))) return rc;
        if ((rc = make_tdnn(h, h->tdnn2[i - 1], p + ".tdnn2", 1))) return rc;
        if ((rc = make_linear(h, h->se1[i - 1], p + ".se_block.conv1.conv.weight", p + ".se_block.conv1.conv.bias"))) return rc;
        if ((rc = make_linear(h, h->se2[i - 1], p + ".se_block.conv2.conv.weight", p + ".se_block.conv2.conv.bias"))) return rc;
        {
            const HostTensor* w2 = getw(h, p + ".se_block.conv2.conv.weight");      // (C, 128, 1)
            std::vector<float> t((size_t)128 * C);
            for (int c = 0; c < C; ++c)
                for (int n = 0; n < 128; ++n) t[(size_t)n * C + c] = w2->data[(size_t)c * 128 + n];
            if ((rc = dev_upload(h, &h->se2T[i - 1], t))) return rc;
            if (h->bf16) {
                const HostTensor* w1 = getw(h, p + ".se_block.conv1.conv.weight");  // (128, C, 1)
                std::vector<uint16_t> b1v((size_t)128 * C), b2v((size_t)128 * C);
                for (size_t k = 0; k < b1v.size(); ++k) { b1v[k] = f32_to_bf16_rne(w1->data[k]); b2v[k] = f32_to_bf16_rne(t[k]); }
                for (int which = 0; which < 2; ++which) {
                    void* d = nullptr;
                    SV_HIP(h, hipMalloc(&d, b1v.size() * 2));
                    h->allocs.push_back(d);
                    SV_HIP(h, hipMemcpy(d, which ? b2v.data() : b1v.data(), b1v.size() * 2, hipMemcpyHostToDevice));
                    (which ? h->se2T_bf[i - 1] : h->se1_bf[i - 1]) = d;
                }
            }
        }
    }
    if ((rc = make_tdnn(h, h->mfa, "mfa", 1))) return rc;
    // asp.tdnn over cat[x, mean, std]: the x columns go through the GEMM, the time-constant columns
    // become a per-utterance bias (ctx) computed by a small linear layer.
    if ((rc = make_conv(h, h->asp_tdnn, "asp.tdnn.conv.conv.weight", "", "asp.tdnn.norm.norm", 1, 0, C3))) return rc;
    if ((rc = make_linear(h, h->asp_ctx, "asp.tdnn.conv.conv.weight", "asp.tdnn.conv.conv.bias", C3, 3 * C3))) return rc;
    if ((rc = make_conv(h, h->asp_conv, "asp.conv.conv.weight", "asp.conv.conv.bias", "", 1))) return rc;
    if ((rc = make_bn(h, "asp_bn.norm", 2 * C3, &h->aspbn_scale, &h->aspbn_shift))) return rc;
    if ((rc = make_linear(h, h->fc, "fc.conv.weight", "fc.conv.bias"))) return rc;
    if (h->cfg.input_norm) {
        const HostTensor *w = getw(h, "instance_norm.weight"), *b = getw(h, "instance_norm.bias");
        if (!w || !b) SV_FAIL(h, SVHIP_ERR_MISSING, "missing instance_norm tensors");
        if ((rc = dev_upload(h, &h->in_w, w->data))) return rc;
        if ((rc = dev_upload(h, &h->in_b, b->data))) return rc;
    }
    // algorithmic FLOPs per utterance: 2 x MACs of every conv / linear (SURVEY §8d counts the same)
    const double T = h->T;
    double f = T * h->blocks0.flops_per_row + T * h->mfa.flops_per_row + T * h->asp_conv.flops_per_row;
    f += T * 2.0 * 128 * (3.0 * C3);                                  // asp.tdnn over the full 9C input, as the reference computes it
    for (int i = 0; i < 3; ++i) {
        f += T * (h->tdnn1[i].flops_per_row + h->tdnn2[i].flops_per_row);
        for (int j = 0; j < 7; ++j) f += T * h->res2[i][j].flops_per_row;
        f += 2.0 * h->se1[i].N * h->se1[i].K + 2.0 * h->se2[i].N * h->se2[i].K;
    }
    f += 2.0 * h->fc.N * h->fc.K;
    h->flops_per_utt = f;
    return SVHIP_OK;
}

int upload_f32(svhip_handle* h, const std::string& name, float** dst) {
    const HostTensor* t = getw(h, name);
    if (!t) SV_FAIL(h, SVHIP_ERR_MISSING, "missing tensor %s", name.c_str());
    return dev_upload(h, dst, t->data);
}

// sinc band-pass filters baked once per weight load (RawNet_baseline.py:313-318,339-357), float32 arithmetic
int bake_sinc(svhip_handle* h) {
    const HostTensor *lo = getw(h, "first_conv.low_hz_"), *bd = getw(h, "first_conv.band_hz_");
    if (!lo || !bd) SV_FAIL(h, SVHIP_ERR_MISSING, "missing sinc parameters");
    const int NF = 128, KS = 251, HALF = 125;
    const float sr = 16000.0f, min_low = 50.0f, min_band = 50.0f;
    const float PI = 3.14159265358979323846f;
    std::vector<float> win(HALF), n_(HALF);
    for (int i = 0; i < HALF; ++i) {
        const float n_lin = (float)(124.5 * i / 124.0);                         // torch.linspace(0, 124.5, 125)
        win[i] = 0.54f - 0.46f * std::cos(2.0f * PI * n_lin / (float)KS);
        n_[i] = 2.0f * PI * (float)(-125 + i) / sr;                             // 2*pi*arange(-125, 0)/16000
    }
    std::vector<float> filt((size_t)NF * KS);
    for (int f = 0; f < NF; ++f) {
        const float low = min_low + std::fabs(lo->data[f]);
        float high = low + min_band + std::fabs(bd->data[f]);
        high = std::fmin(std::fmax(high, min_low), sr / 2);
        const float band = high - low;
        for (int i = 0; i < HALF; ++i) {
            const float left = ((std::sin(high * n_[i]) - std::sin(low * n_[i])) / (n_[i] / 2.0f)) * win[i];
            filt[(size_t)f * KS + i] = left / (2.0f * band);
            filt[(size_t)f * KS + (KS - 1 - i)] = left / (2.0f * band);
        }
        filt[(size_t)f * KS + HALF] = (2.0f * band) / (2.0f * band);
    }
    int rc;
    if (h->bf16) {
        std::vector<uint16_t> pk((size_t)NF * 256, 0);
        for (int f = 0; f < NF; ++f)
            for (int k = 0; k < KS; ++k) pk[(size_t)f * 256 + k] = to_h16(h, filt[(size_t)f * KS + k]);
        uint16_t* d;
        if ((rc = dev_upload(h, &d, pk))) return rc;
        h->rn_filt = d;
        if (h->f16) {
            // the symmetric form (rawnet2.hip, SYM): slot k' = 2 + m carries h[125 + m] (the centre tap halved: its operand is x[c] + x[c]),
            // slots 0 and 1 are zero; right and left halves of a filter are the same numbers by construction (checked here)
            bool symmetric = true;
            for (int f = 0; f < NF && symmetric; ++f)
                for (int i = 0; i < HALF; ++i) symmetric = symmetric && filt[(size_t)f * KS + i] == filt[(size_t)f * KS + (KS - 1 - i)];
            if (symmetric) {
                std::vector<uint16_t> ps((size_t)NF * 128, 0);
                for (int f = 0; f < NF; ++f) {
                    ps[(size_t)f * 128 + 2] = to_h16(h, 0.5f * filt[(size_t)f * KS + HALF]);
                    for (int m = 1; m <= HALF; ++m) ps[(size_t)f * 128 + 2 + m] = to_h16(h, filt[(size_t)f * KS + HALF + m]);
                }
                uint16_t* ds;
                if ((rc = dev_upload(h, &ds, ps))) return rc;
                h->rn_filt_sym = ds;
            }
        }
    } else {
        std::vector<float> pk((size_t)NF * 252, 0.0f);
        for (int f = 0; f < NF; ++f)
            for (int k = 0; k < KS; ++k) pk[(size_t)f * 252 + k] = filt[(size_t)f * KS + k];
        float* d;
        if ((rc = dev_upload(h, &d, pk))) return rc;
        h->rn_filt = d;
        if (h->x3) {        // the split front-end (rn_sinc_x3): hi and lo half planes, k contiguous, zero beyond the 251 taps
            std::vector<uint16_t> pl((size_t)2 * NF * 256, 0);
            for (int f = 0; f < NF; ++f)
                for (int k = 0; k < KS; ++k) {
                    const uint32_t w = x3_split_word(filt[(size_t)f * KS + k]);
                    pl[(size_t)f * 256 + k] = (uint16_t)(w >> 16);
                    pl[(size_t)(NF + f) * 256 + k] = (uint16_t)(w & 0xffffu);
                }
            uint16_t* dx;
            if ((rc = dev_upload(h, &dx, pl))) return rc;
            h->rn_filt_x3 = dx;
        }
    }
    return SVHIP_OK;
}

int finalize_rawnet2(svhip_handle* h) {
    int rc;
    if ((rc = upload_f32(h, "ln.gamma", &h->rn_gamma))) return rc;
    if ((rc = upload_f32(h, "ln.beta", &h->rn_beta))) return rc;
    if ((rc = bake_sinc(h))) return rc;
    if ((rc = make_bn(h, "first_bn", 128, &h->rn_fbn_scale, &h->rn_fbn_shift))) return rc;
    int inpl = 128, bi = 0;
    int T = h->rn_T1;
    double fl = 2.0 * 128 * 251 * (double)(h->cfg.samples - 250);
    for (int li = 0; li < 6; ++li)
        for (int b = 0; b < RN_LAYERS[li]; ++b, ++bi) {
            svhip_handle::RnBlock& B = h->rn_blocks[bi];
            const std::string p = "layer" + std::to_string(li + 1) + "." + std::to_string(b);
            const int planes = RN_FILTERS[li];
            B.cin = inpl; B.cout = planes; B.downsample = (b == RN_LAYERS[li] - 1); B.has_shortcut = inpl != planes;
            if ((rc = make_bn(h, p + ".bn1", inpl, &B.bn1_scale, &B.bn1_shift))) return rc;
            if ((rc = make_conv(h, B.conv1, p + ".conv1.weight", "", p + ".bn2", 1))) return rc;
            if ((rc = make_conv(h, B.conv2, p + ".conv2.weight", "", "", 1))) return rc;
            if (B.has_shortcut && (rc = make_conv(h, B.shortcut, p + ".shortcut.0.weight", "", "", 1))) return rc;
            if (B.has_shortcut && h->bf16 && B.conv2.K % 64 == 0 && inpl % 64 == 0) {
                // conv2 and the shortcut share their output: [conv2 columns (tap-major) | shortcut columns] as one K axis
                const HostTensor* w2 = getw(h, p + ".conv2.weight");           // (planes, planes, 3)
                const HostTensor* ws = getw(h, p + ".shortcut.0.weight");      // (planes, inpl, 1)
                const int K2 = B.conv2.K, Kt = K2 + inpl, Np = B.conv2.Np;
                std::vector<uint16_t> pk((size_t)Np * Kt, 0);
                for (int n = 0; n < planes; ++n) {
                    for (int t = 0; t < 3; ++t)
                        for (int c = 0; c < planes; ++c) pk[(size_t)n * Kt + t * planes + c] = to_h16(h, w2->data[((size_t)n * planes + c) * 3 + t]);
                    for (int c = 0; c < inpl; ++c) pk[(size_t)n * Kt + K2 + c] = to_h16(h, ws->data[(size_t)n * inpl + c]);
                }
                uint16_t* d;
                if ((rc = dev_upload(h, &d, pk))) return rc;
                B.conv2sc_W = d;
            }
            if ((rc = upload_f32(h, p + ".afms.alpha", &B.alpha))) return rc;
            if ((rc = make_linear(h, B.afms_fc, p + ".afms.fc.weight", p + ".afms.fc.bias"))) return rc;
            {
                const HostTensor* fw = getw(h, p + ".afms.fc.weight");                  // (planes, planes)
                std::vector<float> t((size_t)planes * planes);
                for (int n = 0; n < planes; ++n)
                    for (int c = 0; c < planes; ++c) t[(size_t)c * planes + n] = fw->data[(size_t)n * planes + c];
                if ((rc = dev_upload(h, &B.afms_fcT, t))) return rc;
            }
            fl += (double)T * (B.conv1.flops_per_row + B.conv2.flops_per_row + (B.has_shortcut ? B.shortcut.flops_per_row : 0.0));
            fl += 2.0 * planes * planes;
            if (B.downsample) T /= 3;
            inpl = planes;
        }
    if ((rc = make_bn(h, "bn_before_agg", 512, &h->rn_agg_scale, &h->rn_agg_shift))) return rc;
    if ((rc = make_conv(h, h->rn_att0, "attention.0.weight", "attention.0.bias", "attention.2", 1))) return rc;
    if ((rc = make_conv(h, h->rn_att3, "attention.3.weight", "attention.3.bias", "", 1))) return rc;
    if ((rc = make_linear(h, h->rn_fc, "fc.weight", "fc.bias"))) return rc;
    fl += (double)T * (h->rn_att0.flops_per_row + h->rn_att3.flops_per_row) + 2.0 * h->rn_fc.N * h->rn_fc.K;
    h->flops_per_utt = fl;
    return SVHIP_OK;
}

int alloc_workspace(svhip_handle* h) {
    const svhip_config& c = h->cfg;
    const size_t B = c.max_batch, T = h->T, M = B * T, C = c.channels, C3 = 3 * C, e = h->esz;
    int rc;
    if ((rc = dev_alloc(h, &h->d_wav, B * (size_t)c.samples))) return rc;
    if ((rc = dev_alloc(h, &h->d_feat, B * c.n_mels * T))) return rc;
    if ((rc = dev_alloc(h, &h->d_pstats, B * c.n_mels * 2))) return rc;
    if (h->x3 && (rc = dev_alloc(h, &h->d_xscale, 2 * (4 + 256)))) return rc;       // (one set per lane slice)
    if (h->fb.sym_hi) {
        if ((rc = dev_alloc(h, &h->d_logmel, B * c.n_mels * T))) return rc;
        if ((rc = dev_alloc(h, &h->d_fpart, B * c.n_mels * ((T + 63) / 64)))) return rc;
    }
    if ((rc = dev_alloc(h, &h->d_zero, 64))) return rc;
    SV_HIP(h, hipMemset(h->d_zero, 0, 256));
    {
        std::vector<float> one(4096, 1.0f), zero(4096, 0.0f);
        if ((rc = dev_upload(h, &h->d_ones, one))) return rc;
        if ((rc = dev_upload(h, &h->d_zeros, zero))) return rc;
    }
    if ((rc = dev_alloc(h, &h->d_emb, B * (size_t)c.embed_dim))) return rc;
    if ((rc = dev_alloc(h, &h->d_status, 4))) return rc;
    SV_HIP(h, hipMemset(h->d_status, 0, 16));
    SV_HIP(h, hipHostMalloc((void**)&h->host_flag, 64, hipHostMallocMapped));
    *h->host_flag = 0;
    SV_HIP(h, hipHostGetDevicePointer((void**)&h->host_flag_dev, h->host_flag, 0));
    if (c.model == SVHIP_MODEL_RAWNET2) {
        h->rn_T1 = (c.samples - 250) / 3;
        const size_t per_utt = (size_t)h->rn_T1 * 128;           // largest activation: (T1, 128); later stages shrink 3x per doubling
        h->rn_buf_bytes = B * per_utt * e;
        for (int i = 0; i < 6; ++i) {
            char* q;
            if ((rc = dev_alloc(h, &q, B * per_utt * e + 256))) return rc;
            h->rn_buf[i] = q;
            SV_HIP(h, hipMemset(q + h->rn_buf_bytes, 0, 256));          // the zero tail (no kernel writes past the payload)
        }
        if ((rc = dev_alloc(h, &h->rn_stats, B * 2))) return rc;
        if (h->bf16 || h->x3) {                                  // LayerNorm output in 16 bits, zero-tailed rows (operand of the 16-bit / split sinc kernels)
            h->rn_Lp = (int)round_up(c.samples + RN_XN_TAIL, 64);
            uint16_t* q;
            if ((rc = dev_alloc(h, &q, (h->x3 ? 4 : 2) * B * (size_t)h->rn_Lp))) return rc;      // (F32X3: hi and lo parts of both copies)
            h->rn_xn = q;
        }
        if ((rc = dev_alloc(h, &h->rn_part, B * (size_t)(rn_block128_ntiles(h->rn_T1) + 1) * 4 * 128))) return rc;
        if ((rc = dev_alloc(h, &h->rn_mean, B * 512))) return rc;
        if ((rc = dev_alloc(h, &h->rn_scratch, B * 16 * 512))) return rc;
        if ((rc = dev_alloc(h, &h->rn_s, B * 512 * 2))) return rc;
        int tf = h->rn_T1;
        for (int i = 0; i < 6; ++i) tf /= 3;                      // six max_pool1d(3) stages follow the front-end
        if (tf < 1) SV_FAIL(h, SVHIP_ERR_INVALID, "utterance too short for RawNet2 (%d samples)", c.samples);
        if ((rc = dev_alloc(h, &h->rn_logits, B * (size_t)tf * 512))) return rc;
        if ((rc = dev_alloc(h, &h->rn_pooled, B * 1024))) return rc;
        if (h->bf16) {          // K-slice partials of fc (K = 1 024: four slices of 256) at full batches, 16-bit handles
            h->lin_part_per_utt = (size_t)4 * (size_t)std::max(128, c.embed_dim);
            if ((rc = dev_alloc(h, &h->d_lin_part, B * h->lin_part_per_utt))) return rc;
        }
    }
    if (c.model == SVHIP_MODEL_ECAPA) {
        char* p;
        auto actbuf = [&](void** dst, size_t elems) -> int {
            int r = dev_alloc(h, &p, elems * e + 256);
            *dst = p;
            return r;
        };
        if ((rc = actbuf(&h->X_in, M * c.n_mels))) return rc;
        if ((rc = actbuf(&h->X0, M * C))) return rc;
        if ((rc = actbuf(&h->H1, M * C))) return rc;
        if ((rc = actbuf(&h->H2, M * C))) return rc;
        if ((rc = actbuf(&h->H3, M * C))) return rc;
        if ((rc = actbuf(&h->CAT, M * C3))) return rc;
        if ((rc = actbuf(&h->MFA, M * C3))) return rc;
        if ((rc = actbuf(&h->ATT, M * 128))) return rc;
        if ((rc = dev_alloc(h, &h->LOGITS, M * C3))) return rc;
        if ((rc = dev_alloc(h, &h->d_mean, B * C))) return rc;
        if ((rc = dev_alloc(h, &h->d_s1, B * 128))) return rc;
        if ((rc = dev_alloc(h, &h->d_s2, B * C))) return rc;
        if ((rc = dev_alloc(h, &h->d_gstats, B * 2 * C3))) return rc;
        if ((rc = dev_alloc(h, &h->d_ctx, B * 128))) return rc;
        h->lin_part_per_utt = (size_t)((2 * C3 + 383) / 384) * (size_t)std::max(128, c.embed_dim);
        if ((rc = dev_alloc(h, &h->d_lin_part, B * h->lin_part_per_utt))) return rc;
        if ((rc = dev_alloc(h, &h->d_pool_raw, B * 2 * C3))) return rc;
        if ((rc = dev_alloc(h, &h->d_pool_bn, B * 2 * C3))) return rc;
        if (h->x3 && (rc = dev_alloc(h, reinterpret_cast<char**>(&h->s32_buf), M * C3 * 4 + 256))) return rc;
        if (h->x3 && C % 32 == 0 && (rc = dev_alloc(h, reinterpret_cast<char**>(&h->cat_s32), M * C3 * 4 + 256))) return rc;
        if (h->x3 && (C == 512 || C == 1024)) {
            if ((rc = dev_alloc(h, reinterpret_cast<char**>(&h->h2_s32), M * C * 4 + 256))) return rc;
            for (int i = 0; i < 2; ++i) if ((rc = dev_alloc(h, reinterpret_cast<char**>(&h->u_s32[i]), M * (C / 8) * 4 + 256))) return rc;
        }
        h->colsum_region = (int64_t)((M + 255) / 256 + 2) * 16 * C3;
        if ((rc = dev_alloc(h, &h->d_colsum, (size_t)4 * h->colsum_region))) return rc;
    }
    return SVHIP_OK;
}

// zero page of a conv-gather GEMM whose A operand starts at `A`: the zero tail of the RawNet2 activation buffer that holds A (behind
// the operand, within 4 GiB: what gemm_pw3's 16-bit conv-gather form needs), else the handle's stand-alone zero page
const void* zero_page_for(const svhip_handle* h, const void* A) {
    const char* a = static_cast<const char*>(A);
    for (int i = 0; i < 6; ++i) {
        const char* b = static_cast<const char*>(h->rn_buf[i]);
        if (b && a >= b && a < b + h->rn_buf_bytes) return b + h->rn_buf_bytes;
    }
    return h->d_zero;
}

// ---- GEMM call helper -------------------------------------------------------------------------------
int conv_gemm(svhip_handle* h, const char* label, const ConvLayer& L, const void* A, int lda, void* Y, int ldy, int M,
              int act1, int act2 = ACT_NONE, const void* A2 = nullptr, int lda2 = 0, const float* bias_utt = nullptr,
              int ld_bu = 0, bool out_f32 = false, int T = 0, int pad_mode = PAD_REFLECT, const void* R = nullptr, int ldr = 0,
              float* colsum = nullptr, int colsum_sq = 0, int64_t colsum_stride = 0, const void* A_s32 = nullptr, int lda_s32 = 0) {
    GemmParams p;
    p.colsum = colsum; p.colsum_sq = colsum_sq; p.colsum_stride = colsum_stride;
    h->last_colsum_done = false;
    h->side_done = false;
    p.R = R; p.ldr = ldr; p.zero_page = zero_page_for(h, A);
    p.zeros = h->d_zeros; p.ones = h->d_ones; p.cv_off = h->opt.cv_off; p.n128_off = h->opt.n128_off;
    p.A = A; p.A2 = A2; p.W = L.W; p.Y = Y;
    p.bias = L.bias; p.bias_utt = bias_utt; p.scale = L.scale; p.shift = L.shift;
    p.M = M; p.N = L.N; p.K = L.K; p.Kp = L.Kp; p.Wrows = L.Np;
    p.lda = lda; p.lda2 = lda2; p.ldy = ldy; p.ld_bu = ld_bu;
    p.T = T > 0 ? T : h->T; p.taps = L.taps; p.dil = L.dil; p.cin = L.cin; p.pad_mode = pad_mode;
    p.act1 = act1; p.act2 = act2; p.out_f32 = out_f32 ? 1 : 0;
    p.f16 = h->f16 ? 1 : 0; p.pw3_cus = h->opt.pw3_cus; p.tail_split = h->opt.pw3_tail_off ? 0 : 1; p.pw4 = h->opt.pw4;
    const bool bf = h->bf16;
    hipStream_t st = h->cur;
    (void)label;
    p.num_cu = h->num_cu;
    if (h->x3 && L.Ws32 && h->s32_buf && !A2 && !bias_utt && !out_f32 && !R) {
        // the GELU layers of an F32X3 handle on the persistent 256 x 256 kernel: A is split into the S32 layout by one elementwise
        // pass, W was split at load time
        GemmParams q = p;
        q.A = A_s32 ? A_s32 : h->s32_buf; q.lda = A_s32 ? lda_s32 : L.K; q.W = L.Ws32; q.x3 = 2;
        q.side_a = h->side_a; q.side_b = h->side_b; q.side_lda = h->side_lda; q.side_ldb = h->side_ldb; q.side_c = h->side_c;
        if (!gemm_pw3x3_supported(q)) q.side_a = q.side_b = nullptr, q.side_c = 0;
        // (utterances shorter than a tile: no column sums from this kernel — the caller then takes the squeeze / statistics kernels)
        if (!gemm_pw3x3_supported(q) && q.colsum) q.colsum = nullptr;
        h->side_done = q.side_c != 0;
        h->side_c = 0;
        if (gemm_pw3x3_supported(q)) {
            int rc = A_s32 ? SVHIP_OK      // (the producer already wrote the split form: se_apply)
                           : run(h, "split_s32", 0, [&]() { return launch_split_s32(reinterpret_cast<const float*>(A), lda, h->s32_buf, M, L.K, st); });
            if (rc) return rc;
            if (q.colsum) { h->last_colsum_done = true; h->last_colsum_groups = 2; }
            return run(h, "gemm_pw3x3", (double)M * L.flops_per_row, [&]() { return launch_gemm_pw3x3(q, st); });
        }
    }
    if (!A) SV_FAIL(h, SVHIP_ERR_STATE, "%s: the operand exists only in the split layout and the kernel that reads it does not take this shape", label);
    if (h->x3) {              // gemm_pw takes the pre-split weights, the generic kernel (A2 / ragged shapes) the fp32 ones
        p.x3 = 1;
        if (gemm_pw_supported(p, false) && L.Wsplit) p.W = L.Wsplit;
    }
    if (p.colsum) {                       // only the pw2 / pw3 epilogues produce the partials; otherwise the caller falls back
        if (gemm_pw2_supported(p, bf) && p.taps == 1) { h->last_colsum_done = true; h->last_colsum_groups = gemm_colsum_groups(p, bf); }
        else p.colsum = nullptr;
    }
    // profile labels name the kernel instance (one label == one kernel symbol in a rocprofv3 trace)
    const GemmRoute route = gemm_route(p, bf);
    const char* klabel = route == ROUTE_PW3 ? ((p.pw4 && gemm_pw4_supported(p, bf)) ? "gemm_pw4" : "gemm_pw3") : route == ROUTE_PW3CV ? "gemm_pw3cv16" : route == ROUTE_N128 ? "gemm_n128" : route == ROUTE_PW2 ? (L.taps > 1 ? "gemm_pw2_conv" : "gemm_pw2")
                         : L.taps > 1 ? (A2 ? "gemm_conv_add" : "gemm_conv") : (route == ROUTE_GENERIC ? "gemm_generic" : "gemm_pw");
    char shaped[96];
    if (h->opt.layer_labels) {            // developer hook (SVHIP_LAYER_LABELS): one profile row per GEMM shape
        snprintf(shaped, sizeof(shaped), "%s M%d N%d K%d", klabel, M, L.N, L.K);
        klabel = shaped;
    }
    return run(h, klabel, (double)M * L.flops_per_row, [&]() { return launch_gemm(p, bf, st); });
}

inline void* off(void* base, size_t elems, int esz) { return reinterpret_cast<char*>(base) + elems * esz; }
inline const void* off(const void* base, size_t elems, int esz) { return reinterpret_cast<const char*>(base) + elems * esz; }

// ECAPA_TDNN.forward (models/ECAPA_TDNN.py:460-502) on device-resident features (B, n_mels, T)
// for the utterances [b0, b0 + B) of the call, enqueued on h->cur.  Every workspace buffer is frame-major, so a
// batch slice is just a row offset: two slices can run concurrently on two streams (lanes).
int ecapa_forward_part(svhip_handle* h, const float* d_feat_all, int b0, int B) {
    const svhip_config& c = h->cfg;
    const int T = h->T, M = B * T, C = c.channels, C3 = 3 * C, C8 = C / 8, e = h->esz;
    const bool bf = h->bf16;
    hipStream_t st = h->cur;
    const size_t r0 = (size_t)b0 * T;                       // first activation row of the slice
    const float* d_feat = d_feat_all + (size_t)b0 * c.n_mels * T;
    void* X_in = off(h->X_in, r0 * c.n_mels, e);
    void* X0 = off(h->X0, r0 * C, e);
    void* H1 = off(h->H1, r0 * C, e);
    void* H2 = off(h->H2, r0 * C, e);
    void* H3 = off(h->H3, r0 * C, e);
    void* CAT = off(h->CAT, r0 * C3, e);
    void* MFA = off(h->MFA, r0 * C3, e);
    void* ATT = off(h->ATT, r0 * 128, e);
    float* LOGITS = h->LOGITS + r0 * C3;
    float* d_pstats = h->d_pstats + (size_t)b0 * c.n_mels * 2;
    float* d_mean = h->d_mean + (size_t)b0 * C;
    float* d_s1 = h->d_s1 + (size_t)b0 * 128;
    float* d_s2 = h->d_s2 + (size_t)b0 * C;
    float* d_gstats = h->d_gstats + (size_t)b0 * 2 * C3;
    float* d_ctx = h->d_ctx + (size_t)b0 * 128;
    float* d_pool_raw = h->d_pool_raw + (size_t)b0 * 2 * C3;
    float* d_pool_bn = h->d_pool_bn + (size_t)b0 * 2 * C3;
    float* d_emb = h->d_emb + (size_t)b0 * c.embed_dim;
    (void)d_s1;
    float* cs_base = ((bf || h->x3) && h->d_colsum) ? h->d_colsum + (b0 ? 2 * h->colsum_region : 0) : nullptr;
    int rc;
    // F32X3: se_apply also leaves each block output in the S32 split layout (CAT's twin), so tdnn1 of the next block and mfa read
    // their A operand without a conversion pass
    char* cat32 = h->cat_s32 ? static_cast<char*>(h->cat_s32) + r0 * C3 * 4 : nullptr;
    // ... and when every consumer of a block output takes the split operand at this batch size (tdnn1 of the next block, mfa: the
    // persistent X3 kernel; the next se_apply reads its residual as hi + lo), the fp32 copy is not written at all
    auto x3_route = [&](const ConvLayer& L, const void* a32, int lda32, bool cs) {
        if (!h->x3 || !L.Ws32 || !h->s32_buf || !a32) return false;
        GemmParams q;
        q.A = a32; q.lda = lda32; q.W = L.Ws32; q.x3 = 2; q.Y = MFA; q.ldy = L.N;
        q.bias = L.bias; q.scale = L.scale; q.shift = L.shift;
        q.M = M; q.N = L.N; q.K = L.K; q.Kp = L.Kp; q.Wrows = L.Np; q.T = T; q.taps = L.taps; q.act1 = ACT_GELU; q.num_cu = h->num_cu;
        q.pw3_cus = h->opt.pw3_cus; q.tail_split = h->opt.pw3_tail_off ? 0 : 1;
        if (cs) { q.colsum = cs_base; q.colsum_sq = 1; q.colsum_stride = h->colsum_region; }
        return gemm_pw3x3_supported(q);
    };
    const bool s32_only = cat32 && x3_route(h->tdnn1[1], cat32, C3, false) && x3_route(h->tdnn1[2], cat32, C3, false) &&
                          x3_route(h->mfa, cat32, C3, cs_base != nullptr) && !h->opt.x3_keep_f32;
    if (s32_only) h->cat_f32_stale = true;
    bool b0_done = false, x0_s32 = false, b0_cv = false;
    GemmParams q0;
    float* xscale = h->d_xscale ? h->d_xscale + (b0 ? 4 + 256 : 0) : nullptr;
    if (h->x3 && h->blocks0.Wcv && h->s32_buf) {
        // F32X3: blocks.0 on the persistent kernel's conv-gather form: the features go to the S32 layout with rows zero-padded to
        // cv_cin channels (one small pass), the im2col view is formed by the operand DMAs
        const ConvLayer& L = h->blocks0;
        GemmParams& q = q0;
        q.A = h->s32_buf; q.lda = L.cv_cin; q.W = L.Wcv; q.Wrows = L.N; q.x3 = 2; q.Y = X0; q.ldy = C;
        q.bias = L.bias; q.scale = L.scale; q.shift = L.shift;
        q.M = M; q.N = L.N; q.K = L.taps * L.cv_cin; q.Kp = L.cv_Kp; q.T = T; q.taps = L.taps; q.dil = L.dil; q.cin = L.cv_cin; q.pad_mode = PAD_REFLECT;
        q.act1 = ACT_GELU; q.act2 = ACT_NONE; q.num_cu = h->num_cu; q.pw3_cus = h->opt.pw3_cus; q.tail_split = h->opt.pw3_tail_off ? 0 : 1;
        // (with s32_only and tdnn1 of the first block on the X3 kernel, X0 itself is written in the split layout: no conversion pass,
        //  block 1's residual is read as hi + lo, svhip_get_stage rebuilds the fp32 view)
        q.y_s32 = (s32_only && x3_route(h->tdnn1[0], X0, C, false)) ? 1 : 0;
        q.in_scale = xscale;
        b0_cv = gemm_pw3cv_supported(q);
    }
    // the prologue's range guard (F32X3: half-precision planes carry |x| <= 65504): with the scaled first convolution only a non-finite
    // input is reported — a finite one of any magnitude is brought into the planes' range by an exact power of two (round 6)
    if (!h->xin_ready && (rc = run(h, "prologue", 0, [&]() {
             return launch_prologue(d_feat, X_in, bf, B, c.n_mels, T, c.log_input, h->in_w, h->in_b, d_pstats, st,
                                    h->x3 ? h->d_status : nullptr, h->host_flag_dev, (b0_cv && xscale) ? 3.0e38f : 65504.0f);
         }))) return rc;
    if (b0_cv) {
        const ConvLayer& L = h->blocks0;
        if (xscale && (rc = run(h, "in_scale", 0, [&]() {
                 return launch_in_scale(static_cast<const float*>(X_in), (int64_t)M * c.n_mels, reinterpret_cast<uint32_t*>(xscale + 4), xscale, st, L.cv_wscale);
             }))) return rc;
        if ((rc = run(h, "split_s32", 0, [&]() { return launch_split_s32(static_cast<const float*>(X_in), c.n_mels, h->s32_buf, M, L.cv_cin, st, L.cv_cin, c.n_mels, xscale); }))) return rc;
        if ((rc = run(h, "gemm_pw3cv", (double)M * L.flops_per_row, [&]() { return launch_gemm_pw3cv(q0, st); }))) return rc;
        b0_done = true;
        x0_s32 = q0.y_s32 != 0;
    }
    if (!b0_done && (rc = conv_gemm(h, "gemm_blocks0", h->blocks0, X_in, c.n_mels, X0, C, M, ACT_GELU))) return rc;
    h->x0_is_s32 = x0_s32;
    const void* xin = x0_s32 ? nullptr : X0;
    int ldin = C;
    const void* xin32 = x0_s32 ? X0 : nullptr;
    int ldin32 = C;
    for (int i = 0; i < 3; ++i) {
        const void* h2_32 = nullptr;      // F32X3: the chain output in the S32 layout (tdnn2's A operand)
        bool r2_done = false;
        // F32X3: seven launches of gemm_pw3's Res2Net step form; step j reads U_j = c_j + y_{j-1} (S32) and writes y_j (S32, into the
        // chain output) and U_{j+1}; no fp32 copy of the chain exists
        char* h2s = h->h2_s32 ? static_cast<char*>(h->h2_s32) + r0 * C * 4 : nullptr;
        char* us[2] = {h->u_s32[0] ? static_cast<char*>(h->u_s32[0]) + r0 * C8 * 4 : nullptr, h->u_s32[1] ? static_cast<char*>(h->u_s32[1]) + r0 * C8 * 4 : nullptr};
        auto step_params = [&](int j) {
            const ConvLayer& L = h->res2[i][j - 1];
            GemmParams q;
            q.A = us[(j - 1) & 1]; q.lda = C8; q.W = L.Ws32; q.Wrows = L.N; q.x3 = 2;
            q.bias = L.bias; q.scale = L.scale; q.shift = L.shift;
            q.M = M; q.N = L.N; q.K = L.K; q.Kp = L.Kp; q.T = T; q.taps = 3; q.dil = L.dil; q.cin = L.cin; q.pad_mode = PAD_REFLECT;
            q.act1 = ACT_RELU; q.act2 = ACT_NONE; q.num_cu = h->num_cu; q.pw3_cus = h->opt.pw3_cus; q.tail_split = h->opt.pw3_tail_off ? 0 : 1;
            q.Y = h2s + (size_t)j * C8 * 4; q.ldy = C;
            if (j < 7) { q.R = static_cast<const float*>(H1) + (size_t)(j + 1) * C8; q.ldr = C; q.Y2 = us[j & 1]; q.lda2 = C8; }
            return q;
        };
        // (C / 8 = 128: the dedicated 128 x 128 kernel, two workgroups per CU, any batch size; C / 8 = 64, or SVHIP_R2_BIG=1: the R2 form
        //  of the persistent 256 x 256 kernel)
        const bool r2_small = h->x3 && h2s && us[0] && us[1] && h->res2[i][0].Ws32 && !h->opt.r2_big && r2_step_supported(step_params(1)) &&
                              x3_route(h->tdnn2[i], h2s, C, false);      // (tdnn2 must be able to read the chain output in the split layout)
        const bool r2_plan = r2_small || (h->x3 && h2s && us[0] && us[1] && h->res2[i][0].Ws32 && gemm_pw3r2_supported(step_params(1)));
        if (r2_plan) {      // tdnn1 writes the pass-through chunk and the first step's input in the split layout itself (when it takes the X3 kernel)
            h->side_a = h2s; h->side_lda = C; h->side_b = us[0]; h->side_ldb = C8; h->side_c = C8;
        }
        if ((rc = conv_gemm(h, "gemm_tdnn", h->tdnn1[i], (s32_only && (i > 0 || x0_s32)) ? nullptr : xin, ldin, H1, C, M, ACT_GELU, ACT_NONE, nullptr, 0, nullptr, 0,
                            false, 0, PAD_REFLECT, nullptr, 0, nullptr, 0, 0, xin32, ldin32))) return rc;
        const bool side_done = h->side_done;
        h->side_c = 0;
        if (r2_plan) {
            {
                if (!side_done) {
                    if ((rc = run(h, "split_s32", 0, [&]() { return launch_split_s32(static_cast<const float*>(H1), C, h2s, M, C8, st, C); }))) return rc;
                    if ((rc = run(h, "split_s32", 0, [&]() { return launch_split_s32(static_cast<const float*>(H1) + C8, C, us[0], M, C8, st, C8); }))) return rc;
                }
                for (int j = 1; j < 8; ++j) {
                    const GemmParams q = step_params(j);
                    if (r2_small) {
                        if ((rc = run(h, "r2_step", (double)M * h->res2[i][j - 1].flops_per_row, [&]() { return launch_r2_step(q, st); }))) return rc;
                    } else if ((rc = run(h, "gemm_pw3r2", (double)M * h->res2[i][j - 1].flops_per_row, [&]() { return launch_gemm_pw3r2(q, st); }))) return rc;
                }
                h2_32 = h2s;
                r2_done = true;
            }
        }
        if (r2_done) {
        } else if (bf && res2net_chain_supported(C, T, h->res2[i][0].dil, h->res2[i][0].Kp)) {
            Res2Params rp;
            rp.H1 = H1; rp.H2 = H2; rp.ld = C; rp.T = T; rp.dil = h->res2[i][0].dil; rp.Kp = h->res2[i][0].Kp;
            // small batches (the reference's per-file calls: B = num_eval crops): time slices, so that the chip is not left to B workgroups
            rp.slices = h->opt.r2_slices >= 0 ? std::max(1, h->opt.r2_slices) : res2net_chain_slices(B, C, T, rp.dil, h->num_cu);
            double fl = 0;
            for (int j = 0; j < 7; ++j) {
                rp.W[j] = h->res2[i][j].W; rp.bias[j] = h->res2[i][j].bias;
                rp.scale[j] = h->res2[i][j].scale; rp.shift[j] = h->res2[i][j].shift;
                fl += (double)M * h->res2[i][j].flops_per_row;
            }
            if ((rc = run(h, rp.slices > 1 ? "res2net_slices" : "res2net_chain", fl, [&]() { return launch_res2net_chain(rp, B, C, st); }))) return rc;
        } else {
            if ((rc = run(h, "copy_cols", 0, [&]() { return launch_copy_cols(H1, C, H2, C, bf, M, C8, st); }))) return rc;
            for (int j = 1; j < 8; ++j) {
                const void* a = off(H1, (size_t)j * C8, e);
                const void* a2 = j >= 2 ? off(H2, (size_t)(j - 1) * C8, e) : nullptr;
                if ((rc = conv_gemm(h, "gemm_res2net", h->res2[i][j - 1], a, C, off(H2, (size_t)j * C8, e), C, M, ACT_RELU,
                                    ACT_NONE, a2, C)))
                    return rc;
            }
        }
        // tdnn2; its epilogue also leaves per-utterance column sums (the SE squeeze) when the pw2 kernel runs
        if ((rc = conv_gemm(h, "gemm_tdnn", h->tdnn2[i], r2_done ? nullptr : H2, C, H3, C, M, ACT_GELU, ACT_NONE, nullptr, 0, nullptr, 0, false, 0,
                            PAD_REFLECT, nullptr, 0, cs_base, 0, h->colsum_region, h2_32, C))) return rc;
        const bool from_part = h->last_colsum_done;      // the squeeze comes straight from the GEMM's column-sum partials
        if (!from_part) {
            if ((rc = run(h, "se_mean", 0, [&]() { return launch_colmean(H3, bf, C, B, T, C, d_mean, st); }))) return rc;
        }
        if ((rc = run(h, "se_mlp", 4.0 * B * 128 * C, [&]() {
                 return launch_se_mlp(from_part ? nullptr : d_mean, from_part ? cs_base : nullptr, T,
                                      bf ? (const void*)h->se1_bf[i] : (const void*)h->se1[i].W, h->se1[i].bias,
                                      bf ? (const void*)h->se2T_bf[i] : (const void*)h->se2T[i], h->se2[i].bias, d_s2, bf, B, C, 128, st,
                                      h->last_colsum_groups);
             }))) return rc;
        void* xout = off(CAT, (size_t)i * C, e);
        void* xout32 = cat32 ? cat32 + (size_t)i * C * 4 : nullptr;
        if ((rc = run(h, "se_apply", 0, [&]() {
                 return launch_se_apply(H3, C, d_s2, xin, ldin, s32_only ? nullptr : xout, C3, bf, B, T, C, st, xout32, C3,
                                        s32_only && (i > 0 || x0_s32) ? xin32 : nullptr, ldin32);
             })))
            return rc;
        xin = xout;
        xin32 = xout32;
        ldin = C3;
        ldin32 = C3;
    }
    if ((rc = conv_gemm(h, "gemm_mfa", h->mfa, s32_only ? nullptr : CAT, C3, MFA, C3, M, ACT_GELU, ACT_NONE, nullptr, 0, nullptr, 0, false, 0,
                        PAD_REFLECT, nullptr, 0, cs_base, 1, h->colsum_region, cat32, C3))) return rc;
    if (h->last_colsum_done) {
        if ((rc = run(h, "colsum_finalize", 0, [&]() { return launch_colsum_finalize(cs_base, h->colsum_region, true, B, T, C3, M, d_gstats, 1e-12f, st, h->last_colsum_groups); }))) return rc;
    } else {
        if ((rc = run(h, "asp_gstats", 0, [&]() { return launch_colstats(MFA, bf, C3, B, T, C3, d_gstats, 1e-12f, st); }))) return rc;
    }
    if ((rc = run(h, "asp_ctx", 2.0 * B * 128 * 2 * C3, [&]() {
             return launch_rowvec_linear(d_gstats, 2 * C3, h->asp_ctx.W, h->asp_ctx.bias, d_ctx, 128, B, 128, 2 * C3, ACT_NONE, st, h->d_lin_part + (size_t)b0 * h->lin_part_per_utt);
         }))) return rc;
    if ((rc = conv_gemm(h, "gemm_asp_tdnn", h->asp_tdnn, MFA, C3, ATT, 128, M, ACT_RELU, ACT_TANH, nullptr, 0, d_ctx, 128)))
        return rc;
    // bf16: 16 waves per CU, lane-local online softmax (asp_x3.hip's bf16 form: 0.195 against 0.264 ms at B = 256, any T); the
    // one-wave-per-SIMD kernel keeps the channel counts that are not multiples of 256 (and SVHIP_ASP_V1=1: the tests compare the two)
    const bool asp_v2 = bf && C3 % 256 == 0 && h->asp_tdnn.N == 128 && h->asp_conv.Kp == 128 && !h->opt.asp_v1;
    if (asp_v2 || (bf && asp_fused_supported(T, C3, h->asp_tdnn.N, h->asp_conv.Kp))) {
        AspFusedParams ap;
        ap.att = ATT; ap.W = h->asp_conv.W; ap.Kp = h->asp_conv.Kp; ap.bias = h->asp_conv.bias;
        ap.X = MFA; ap.ldx = C3; ap.T = T; ap.C = C3;
        ap.bn_scale = h->aspbn_scale; ap.bn_shift = h->aspbn_shift;
        ap.pooled_raw = d_pool_raw; ap.pooled_bn = d_pool_bn; ap.eps = 1e-12f;
        if (asp_v2) {
            if ((rc = run(h, "asp_bf16", (double)M * h->asp_conv.flops_per_row, [&]() { return launch_asp_bf16(ap, d_gstats, 2 * C3, B, st); }))) return rc;
        } else
        if ((rc = run(h, "asp_fused", (double)M * h->asp_conv.flops_per_row, [&]() { return launch_asp_fused(ap, B, st); }))) return rc;
    } else if (h->x3 && h->asp_conv.Ws32 && asp_x3_supported(T, C3, h->asp_tdnn.N, h->asp_conv.K)) {
        AspX3Params ap;
        ap.att = (const float*)ATT; ap.Ws32 = h->asp_conv.Ws32; ap.X = (const float*)MFA; ap.ldx = C3; ap.T = T; ap.C = C3;
        ap.mref = d_gstats; ap.mref_ld = 2 * C3;                  // [mean | std] per utterance: the means
        ap.bn_scale = h->aspbn_scale; ap.bn_shift = h->aspbn_shift;
        ap.pooled_raw = d_pool_raw; ap.pooled_bn = d_pool_bn; ap.eps = 1e-12f;
        if ((rc = run(h, "asp_x3", (double)M * h->asp_conv.flops_per_row, [&]() { return launch_asp_x3(ap, B, st); }))) return rc;
    } else {
        if ((rc = conv_gemm(h, "gemm_asp_conv", h->asp_conv, ATT, 128, LOGITS, C3, M, ACT_NONE, ACT_NONE, nullptr, 0, nullptr, 0, true)))
            return rc;
        if ((rc = run(h, "asp_pool", 0, [&]() {
                 return launch_asp_pool(LOGITS, MFA, bf, C3, B, T, C3, h->aspbn_scale, h->aspbn_shift, d_pool_raw, d_pool_bn, 1e-12f, st);
             }))) return rc;
    }
    if ((rc = run(h, "fc", 2.0 * B * h->fc.N * h->fc.K, [&]() {
             return launch_rowvec_linear(d_pool_bn, 2 * C3, h->fc.W, h->fc.bias, d_emb, c.embed_dim, B, c.embed_dim, 2 * C3, ACT_NONE, st, h->d_lin_part + (size_t)b0 * h->lin_part_per_utt);
         }))) return rc;
    return SVHIP_OK;
}

// whole batch: one lane, or two half-batches on two streams so that kernel tails, launch gaps and the
// small latency-bound kernels of one half overlap the big GEMMs of the other
int ecapa_forward(svhip_handle* h, const float* d_feat, int B) {
    int rc = SVHIP_OK;
    h->cat_f32_stale = false;
    if (h->lanes == 2 && B >= 64 && !h->x3) {      // (F32X3: the lanes would share the split-operand staging buffer)
        const int B0 = (B / 2 + 3) & ~3;
        SV_HIP(h, hipEventRecord(h->lane_ev[4], h->stream));
        for (int l = 0; l < 2 && !rc; ++l) {
            SV_HIP(h, hipStreamWaitEvent(h->lane_stream[l], h->lane_ev[4], 0));
            h->cur = h->lane_stream[l];
            rc = ecapa_forward_part(h, d_feat, l == 0 ? 0 : B0, l == 0 ? B0 : B - B0);
            h->cur = h->stream;
            if (rc) break;
            SV_HIP(h, hipEventRecord(h->lane_ev[l], h->lane_stream[l]));
            SV_HIP(h, hipStreamWaitEvent(h->stream, h->lane_ev[l], 0));
        }
    } else {
        h->cur = h->stream;
        rc = ecapa_forward_part(h, d_feat, 0, B);
    }
    if (!rc) h->lastB = B;
    return rc;
}

// conv2 + 1 x 1 shortcut of a RawNet2 block as ONE conv-gather GEMM: K = 3 * cout conv columns of hb, then cin columns of `pre`
static GemmParams conv2sc_params(svhip_handle* h, const svhip_handle::RnBlock& K, const void* pre, const void* hb, void* o, int M, int T) {
    GemmParams p;
    p.A = hb; p.W = K.conv2sc_W; p.Y = o; p.zero_page = zero_page_for(h, hb);
    p.zeros = h->d_zeros; p.ones = h->d_ones; p.cv_off = h->opt.cv_off;
    p.M = M; p.N = K.cout; p.K = K.conv2.K; p.Kp = K.conv2.K + K.cin; p.Wrows = K.conv2.Np;
    p.lda = K.cout; p.ldy = K.cout; p.T = T; p.taps = 3; p.dil = 1; p.cin = K.cout; p.pad_mode = PAD_ZERO;
    p.A3 = pre; p.lda3 = K.cin; p.K3 = K.cin;
    p.num_cu = h->num_cu; p.f16 = h->f16 ? 1 : 0; p.pw3_cus = h->opt.pw3_cus; p.tail_split = h->opt.pw3_tail_off ? 0 : 1;
    return p;
}
static bool conv2sc_fits(svhip_handle* h, const svhip_handle::RnBlock& K, const void* pre, const void* hb, void* o, int M, int T) {
    const GemmParams p = conv2sc_params(h, K, pre, hb, o, M, T);
    return gemm_pw2_supported(p, true) && gemm_route(p, true) == ROUTE_PW2;
}

// would conv_gemm route this residual-free convolution to the persistent conv-gather kernel?
static bool conv_cv_persistent(svhip_handle* h, const ConvLayer& L, const void* A, int lda, int M, int T, int pad_mode) {
    GemmParams p;
    p.A = A; p.W = L.W; p.Y = h->d_emb;      // (Y: any 16-byte aligned pointer; the route does not depend on it)
    p.bias = L.bias; p.scale = L.scale; p.shift = L.shift; p.zeros = h->d_zeros; p.ones = h->d_ones; p.zero_page = zero_page_for(h, A);
    p.M = M; p.N = L.N; p.K = L.K; p.Kp = L.Kp; p.Wrows = L.Np; p.lda = lda; p.ldy = L.N; p.T = T;
    p.taps = L.taps; p.dil = L.dil; p.cin = L.cin; p.pad_mode = pad_mode;
    p.num_cu = h->num_cu; p.f16 = h->f16 ? 1 : 0; p.pw3_cus = h->opt.pw3_cus; p.tail_split = h->opt.pw3_tail_off ? 0 : 1; p.cv_off = h->opt.cv_off;
    return h->bf16 && gemm_route(p, true) == ROUTE_PW3CV;
}

// RawNet2.forward (models/RawNet2_custom.py:161-227) on device-resident waveforms (B, L), utterances [b0, b0 + B) of the call,
// enqueued on h->cur.  Every workspace buffer is per-utterance contiguous, so a batch slice is an offset into each.
int rawnet2_forward_part(svhip_handle* h, const float* d_wav_all, int b0, int B) {
    const svhip_config& c = h->cfg;
    const bool bf = h->bf16;
    hipStream_t st = h->cur;
    const int L = c.samples, e = h->esz;
    const size_t per_utt = (size_t)h->rn_T1 * 128;                 // elements of the largest activation of one utterance
    const float* d_wav = d_wav_all + (size_t)b0 * L;
    float* rn_stats = h->rn_stats + (size_t)b0 * 2;
    float* rn_mean = h->rn_mean + (size_t)b0 * 512;
    float* rn_scratch = h->rn_scratch + (size_t)b0 * 16 * 512;
    float* rn_part = h->rn_part + (size_t)b0 * (rn_block128_ntiles(h->rn_T1) + 1) * 4 * 128;
    float* rn_gate[2] = {h->rn_s + (size_t)b0 * 512, h->rn_s + ((size_t)c.max_batch + b0) * 512};
    float* rn_pooled = h->rn_pooled + (size_t)b0 * 1024;
    float* d_emb = h->d_emb + (size_t)b0 * c.embed_dim;
    int rc;
    const bool sinc_x3 = h->x3 && h->rn_filt_x3 && !h->opt.rn_sinc_f32;            // F32X3: the front-end on three fp16 MFMAs per product
    void* rn_xn = bf ? static_cast<char*>(h->rn_xn) + (size_t)b0 * 2 * h->rn_Lp * 2 : sinc_x3 ? static_cast<char*>(h->rn_xn) + (size_t)b0 * 4 * h->rn_Lp * 2 : nullptr;
    const int dt = h->dt;
    if ((rc = run(h, "rn_ln_stats", 0, [&]() { return launch_rn_ln_stats(d_wav, B, L, rn_stats, st, rn_xn, h->rn_Lp, h->rn_gamma, h->rn_beta, dt, sinc_x3); }))) return rc;
    int T = h->rn_T1;
    void *x = off(h->rn_buf[0], b0 * per_utt, e), *pre = off(h->rn_buf[1], b0 * per_utt, e), *hb = off(h->rn_buf[2], b0 * per_utt, e),
         *o = off(h->rn_buf[3], b0 * per_utt, e), *sc = off(h->rn_buf[4], b0 * per_utt, e), *xn = off(h->rn_buf[5], b0 * per_utt, e);
    // developer hook (option rn_stop): return after N residual blocks (0: after the front-end) with x exposed as stage "rn_x"; the
    // unfused kernel sequence runs, whose storage points are those of the fused kernels
    const int stop_after = h->opt.rn_stop;
    // developer hook (tests): SVHIP_RN_SNAP=2 keeps a copy of lrelu(bn1(x)) as block 2 will read it — the first tensor that both the
    // fused 128-channel blocks and the separate kernel sequence materialise — as stage "rn_snap"
    const int snap_at = h->opt.rn_snap;
    auto snapshot = [&](const void* src, int Tn, int Cn) -> int {
        const size_t bytes = (size_t)B * Tn * Cn * e;
        if (h->rn_snap_cap < bytes) {
            void* q = nullptr;
            SV_HIP(h, hipMalloc(&q, bytes));
            h->allocs.push_back(q);
            h->rn_snap = q; h->rn_snap_cap = bytes;
        }
        SV_HIP(h, hipMemcpyAsync(h->rn_snap, src, bytes, hipMemcpyDeviceToDevice, st));
        h->rn_snap_T = Tn; h->rn_snap_C = Cn;
        return SVHIP_OK;
    };
    // F32X3: will block `bn` (entered with Tn frames) run its convolutions on the 128 x 128 split kernel (r2_step.hip modes 1 / 2)?  Its
    // producer then writes lrelu(bn1(x)) straight in the S32 layout (pre_is_s32) instead of fp32
    auto x3_step_block = [&](int bn, int Tn) {
        if (bn > 7 || !h->x3 || h->opt.rn_step_off || stop_after >= 0 || snap_at >= 0) return false;
        const svhip_handle::RnBlock& Kn = h->rn_blocks[bn];
        return Kn.cin % 32 == 0 && Kn.cout % 128 == 0 && Kn.conv1.Ws32 && Kn.conv2.Ws32 && (!Kn.has_shortcut || Kn.shortcut.Ws32) && Tn >= 2;
    };
    bool pre_is_s32 = false;
    // (the split front-end writes block 0's pre-activation itself, in the S32 layout, when block 0 runs on the split convolution kernel)
    const bool sinc_pre = sinc_x3 && x3_step_block(0, T);
    if (sinc_pre) pre_is_s32 = true;
    if ((rc = run(h, "rn_sinc", 2.0 * B * 128.0 * 251.0 * (L - 250), [&]() {
             // (the kernel can also write block 0's pre-activation, but its 8-byte scattered stores make that as dear as the
             //  separate coalesced rn_bn_act pass: measured 0.85 + 0.29 ms either way)
             if (sinc_x3) return launch_rn_sinc_x3(h->rn_filt_x3, h->rn_fbn_scale, h->rn_fbn_shift, reinterpret_cast<float*>(x), B, L, T, rn_xn, h->rn_Lp, h->num_cu, st,
                                                   sinc_pre ? pre : nullptr, h->rn_blocks[0].bn1_scale, h->rn_blocks[0].bn1_shift);
             // fp16 handles: the symmetric form of the sinc convolution (K = 126 instead of 251; option rn_sinc_full keeps round 5's kernel)
             const bool sym = h->f16 && h->rn_filt_sym && !h->opt.rn_sinc_full;
             return launch_rn_sinc(d_wav, rn_stats, h->rn_gamma, h->rn_beta, sym ? h->rn_filt_sym : h->rn_filt, h->rn_fbn_scale, h->rn_fbn_shift, x, dt, B, L, T, st,
                                   nullptr, nullptr, nullptr, rn_xn, h->rn_Lp, h->num_cu, sym);
         }))) return rc;
    h->rn_dbg_x = x; h->rn_dbg_T = T; h->rn_dbg_C = 128;
    if (stop_after == 0) return SVHIP_OK;
    // bf16: the 128 -> 128 pooled blocks (layer1, layer2) each run as ONE fused kernel + the AFMS gate kernel; the gate of
    // block i is applied by block i + 1 on the way in (or by the rn_afms_apply pass in front of the first GEMM block)
    int first = 0;
    const bool fuse_ok = bf && stop_after < 0 && !h->opt.rn_unfused;
    const bool no_tail = h->opt.rn_unfused != 0;                        // (tests: the separate passes against the fused tail)
    const float *g_alpha = nullptr, *g_gate = nullptr;          // pending gate of the previous fused block
    const void* xin = x;
    for (; fuse_ok && first < 8; ++first) {
        svhip_handle::RnBlock& K = h->rn_blocks[first];
        if (!rn_block128_supported(K.cin, K.cout, T, K.downsample, K.has_shortcut, K.conv1.Kp, K.conv2.Kp)) break;
        RnBlock128Params bp;
        bp.xin = reinterpret_cast<const bf16_t*>(xin);
        bp.alpha = g_alpha; bp.gate = g_gate;
        bp.bn1_scale = K.bn1_scale; bp.bn1_shift = K.bn1_shift;
        bp.W1 = reinterpret_cast<const bf16_t*>(K.conv1.W); bp.bn2_scale = K.conv1.scale; bp.bn2_shift = K.conv1.shift;
        bp.W2 = reinterpret_cast<const bf16_t*>(K.conv2.W);
        void* dst = (first & 1) ? hb : o;                        // ping-pong: never the buffer being read
        bp.opool = reinterpret_cast<bf16_t*>(dst);
        bp.colsum = rn_part;
        bp.B = B; bp.T = T; bp.Tout = T / 3; bp.ntiles = rn_block128_ntiles(T); bp.f16 = h->f16 ? 1 : 0;
        const double fl = (double)B * T * (K.conv1.flops_per_row + K.conv2.flops_per_row);
        if ((rc = run(h, "rn_block128", fl, [&]() { return launch_rn_block128(bp, h->num_cu, st); }))) return rc;
        float* gate = rn_gate[first & 1];                        // two gate buffers: block i + 1 reads i's while writing its own
        if ((rc = run(h, "rn_afms_gate", 2.0 * B * K.cout * K.cout, [&]() {
                 return launch_rn_afms_gate(rn_part, rn_block128_nparts(B, bp.T, h->num_cu), B, K.cout, bp.Tout, K.afms_fcT, K.afms_fc.bias, gate, st);
             }))) return rc;
        T /= 3;
        xin = dst;
        g_alpha = K.alpha; g_gate = gate;
    }
    if (first > 0) {
        // x = (o + alpha) * gate and, in the same pass, the next consumer's lrelu(bn(x))
        svhip_handle::RnBlock& Kp = h->rn_blocks[first - 1];
        const float* nsc = first < 8 ? h->rn_blocks[first].bn1_scale : h->rn_agg_scale;
        const float* nsh = first < 8 ? h->rn_blocks[first].bn1_shift : h->rn_agg_shift;
        // x itself is read only as an identity shortcut (or as a debug stage): not written when the next block projects its input
        void* xdst = (first < 8 && h->rn_blocks[first].has_shortcut && stop_after < 0) ? nullptr : x;
        if ((rc = run(h, "rn_afms_apply", 0, [&]() { return launch_rn_afms_apply(xin, xdst, dt, Kp.alpha, g_gate, B, T, Kp.cout, st, nsc, nsh, pre, 0.3f); }))) return rc;
        h->rn_dbg_x = x; h->rn_dbg_T = T; h->rn_dbg_C = Kp.cout;
        if (snap_at == first && b0 == 0 && (rc = snapshot(pre, T, Kp.cout))) return rc;
    }
    for (int bi = first; bi < 8; ++bi) {
        svhip_handle::RnBlock& K = h->rn_blocks[bi];
        const int M = B * T;
        // out = lrelu(bn1(x))                                                         RawNet_baseline.py:222
        // (blocks 1..7 get it from the previous block's AFMS pass, which writes x and lrelu(bn1(x)) together)
        if (((bi == 0 && first == 0) || stop_after >= 0) && !(bi == 0 && sinc_pre)) {
            pre_is_s32 = x3_step_block(bi, T);
            if ((rc = run(h, "rn_bn_act", 0, [&]() { return launch_rn_bn_act(x, pre, dt, K.bn1_scale, K.bn1_shift, M, K.cin, 0.3f, st, pre_is_s32); }))) return rc;
        }
        // conv1 -> bn2 -> lrelu (epilogue), conv2 + shortcut                            :224-226
        // A 1 x 1 shortcut rides in conv2's GEMM as extra K columns when the 256 x 256 kernel takes it (no shortcut tensor in HBM)
        const bool fold_sc = K.has_shortcut && K.conv2sc_W && !no_tail && conv2sc_fits(h, K, pre, hb, o, M, T);
        const void* resid = x;                                                       // identity shortcut takes the pre-BN x (:223)
        const void* resid_in_tail = nullptr;
        const bool tail_fused = !no_tail && rn_tail_supported(dt, K.downsample ? T / 3 : T, K.cout);
        // F32X3: the block's convolutions (and its projection shortcut) on the 128 x 128 split kernel (r2_step.hip, modes 1 / 2) — pre in the
        // S32 layout, conv1's output stays S32 (conv2's operand), conv2 adds the shortcut on the way out.  (Otherwise they run on the tiled
        // kernel that splits its fp32 operands in registers: 170 - 190 TFLOP/s.)
        bool pooled_by_conv = false;
        const bool x3_step = h->x3 && !h->opt.rn_step_off && K.cin % 32 == 0 && K.cout % 128 == 0 && K.conv1.Ws32 && K.conv2.Ws32 &&
                             (!K.has_shortcut || K.shortcut.Ws32) && T >= 2;
        if (K.has_shortcut && !fold_sc && !x3_step) {
            if ((rc = conv_gemm(h, "rn_gemm", K.shortcut, pre, K.cin, sc, K.cout, M, ACT_NONE))) return rc;
            resid = sc;
        }
        if (pre_is_s32 && !x3_step) SV_FAIL(h, SVHIP_ERR_STATE, "RawNet2 block %d: split pre-activation without the split convolution route", bi);
        if (x3_step) {
            GemmParams q1;                                   // conv1: pre (S32) -> lrelu(bn2(.)) in S32
            void* const split_dst = K.has_shortcut ? xn : sc;      // (fp32 pre: its S32 copy goes to a buffer that is free here — the next-x buffer when `sc` holds the projected shortcut)
            q1.A = pre_is_s32 ? pre : split_dst; q1.lda = K.cin; q1.W = K.conv1.Ws32; q1.Wrows = K.cout; q1.x3 = 2;
            q1.scale = K.conv1.scale; q1.shift = K.conv1.shift;
            q1.M = M; q1.N = K.cout; q1.K = 3 * K.cin; q1.Kp = q1.K; q1.T = T; q1.taps = 3; q1.dil = 1; q1.cin = K.cin; q1.pad_mode = PAD_ZERO;
            q1.zero_page = h->d_zeros; q1.Y = hb; q1.ldy = K.cout; q1.num_cu = h->num_cu;
            GemmParams q2 = q1;                              // conv2: h (S32) -> fp32, + the shortcut (identity x, or the projected one)
            q2.A = hb; q2.lda = K.cout; q2.cin = K.cout; q2.K = 3 * K.cout; q2.Kp = q2.K; q2.W = K.conv2.Ws32; q2.scale = nullptr; q2.shift = nullptr;
            q2.Y = o; q2.out_f32 = 1; q2.R = reinterpret_cast<const float*>(K.has_shortcut ? sc : x); q2.ldr = K.cout;
            GemmParams q0 = q1;                              // projection shortcut (k = 1) of pre -> fp32, into the spare activation buffer
            q0.W = K.shortcut.Ws32; q0.taps = 1; q0.K = K.cin; q0.Kp = K.cin; q0.scale = nullptr; q0.shift = nullptr; q0.Y = sc; q0.out_f32 = 1;
            const bool ok = rn_step_supported(q1, 1) && rn_step_supported(q2, 2) && (!K.has_shortcut || rn_step_supported(q0, 2));
            if (ok) {
                if (!pre_is_s32 && (rc = run(h, "split_s32", 0, [&]() { return launch_split_s32(reinterpret_cast<const float*>(pre), K.cin, split_dst, M, K.cin, st); }))) return rc;
                if (K.has_shortcut && (rc = run(h, "rn_step", (double)M * K.shortcut.flops_per_row, [&]() { return launch_rn_step(q0, 2, st); }))) return rc;
                if ((rc = run(h, "rn_step", (double)M * K.conv1.flops_per_row, [&]() { return launch_rn_step(q1, 1, st); }))) return rc;
                // (a pooled block whose tail is not the fused kernel — the long utterances of layers 1 - 3: conv2 pools on its way out)
                pooled_by_conv = K.downsample && !tail_fused && T >= 3 && !h->opt.rn_pool_off && rn_step_supported(q2, 3);
                if ((rc = run(h, "rn_step", (double)M * K.conv2.flops_per_row, [&]() { return launch_rn_step(q2, pooled_by_conv ? 3 : 2, st); }))) return rc;
                goto convs_done;
            }
            if (pre_is_s32) SV_FAIL(h, SVHIP_ERR_STATE, "RawNet2 block %d: the split convolution kernel refused a shape its producer was told it takes", bi);
            if (K.has_shortcut && !fold_sc) {               // (the tiled route after all: its projection shortcut)
                if ((rc = conv_gemm(h, "rn_gemm", K.shortcut, pre, K.cin, sc, K.cout, M, ACT_NONE))) return rc;
                resid = sc;
            }
        }
        if ((rc = conv_gemm(h, "rn_gemm", K.conv1, pre, K.cin, hb, K.cout, M, ACT_NONE, ACT_LRELU03, nullptr, 0, nullptr, 0, false, T, PAD_ZERO))) return rc;
        if (fold_sc) {
            GemmParams p = conv2sc_params(h, K, pre, hb, o, M, T);
            const char* lbl = "gemm_pw2_conv";
            char shaped2[96];
            if (h->opt.layer_labels) { snprintf(shaped2, sizeof(shaped2), "%s M%d N%d K%d+%d", lbl, M, K.cout, K.conv2.K, K.cin); lbl = shaped2; }
            if ((rc = run(h, lbl, (double)M * (K.conv2.flops_per_row + K.shortcut.flops_per_row), [&]() { return launch_gemm(p, true, st); }))) return rc;
        } else {
            // identity shortcut: with the fused block tail and conv2 on the persistent conv-gather kernel (which has no residual
            // operand) the tail adds the block input; otherwise conv2's epilogue does
            if (!K.has_shortcut && tail_fused && bf && conv_cv_persistent(h, K.conv2, hb, K.cout, M, T, PAD_ZERO)) { resid_in_tail = x; resid = nullptr; }
            if ((rc = conv_gemm(h, "rn_gemm", K.conv2, hb, K.cout, o, K.cout, M, ACT_NONE, ACT_NONE, nullptr, 0, nullptr, 0, false, T, PAD_ZERO, resid, resid ? K.cout : 0))) return rc;
        }
    convs_done:
        // AFMS gate; the same pass writes the next consumer's lrelu(bn(.)): block bi+1's bn1, or the aggregation BN after block 7
        const float* nsc = bi < 7 ? h->rn_blocks[bi + 1].bn1_scale : h->rn_agg_scale;
        const float* nsh = bi < 7 ? h->rn_blocks[bi + 1].bn1_shift : h->rn_agg_shift;
        void* npre = stop_after >= 0 ? nullptr : pre;           // (the developer hook keeps the unfused sequence)
        // the block output itself is read only by an identity shortcut of the next block (or as a debug stage)
        const bool x_dead = stop_after < 0 && npre && (bi == 7 || h->rn_blocks[bi + 1].has_shortcut);
        const int Tn = K.downsample ? T / 3 : T;
        if (tail_fused) {
            // max-pool + AFMS + next pre-activation in one launch, the pooled activation held in registers      :228-229, :62-68
            const bool tail_s32 = npre && x3_step_block(bi + 1, Tn);      // (F32X3: the next block's operand straight in the S32 layout)
            char tl[48] = "rn_tail";
            if (h->opt.layer_labels) snprintf(tl, sizeof(tl), "rn_tail T%d C%d", T, K.cout);
            if ((rc = run(h, tl, 2.0 * B * K.cout * K.cout, [&]() {
                     // (small batches: slice sums in rn_scratch, the gate in rn_gate[0]; option rn_tail_big keeps one workgroup per utterance)
                     const bool sliced = !h->opt.rn_tail_big;
                     return launch_rn_tail(o, x_dead ? nullptr : xn, npre, dt, K.downsample, K.alpha, K.afms_fcT, K.afms_fc.bias, nsc, nsh, B, T, K.cout, 0.3f, st,
                                           resid_in_tail, sliced ? rn_scratch : nullptr, sliced ? rn_gate[0] : nullptr, h->num_cu, tail_s32);
                 }))) return rc;
            pre_is_s32 = tail_s32;
            T = Tn;
        } else {
            void* y = o;
            if (K.downsample && pooled_by_conv) {                                        // (F32X3: conv2 pooled on its way out, into o)
                T /= 3;
            } else if (K.downsample) {                                                   // :228-229
                if ((rc = run(h, "rn_maxpool3", 0, [&]() { return launch_rn_maxpool3(o, hb, dt, B, T, K.cout, st); }))) return rc;
                T /= 3;
                y = hb;
            }
            // AFMS: (y + alpha) * sigmoid(fc(mean_t y))                                     :62-68
            if ((rc = run(h, "rn_afms_mean", 0, [&]() { return launch_colmean(y, dt, K.cout, B, T, K.cout, rn_mean, st, rn_scratch, 16); }))) return rc;
            if ((rc = run(h, "rn_afms_gate", 2.0 * B * K.cout * K.cout, [&]() {
                     return launch_rn_afms_gate(rn_mean, 1, B, K.cout, 1, K.afms_fcT, K.afms_fc.bias, rn_gate[0], st);
                 }))) return rc;
            // (F32X3: when the next block runs on the split convolution kernel its pre-activation is written in the S32 layout right here)
            const bool next_s32 = npre && x3_step_block(bi + 1, T);
            if ((rc = run(h, "rn_afms_apply", 0, [&]() { return launch_rn_afms_apply(y, x_dead ? nullptr : xn, dt, K.alpha, rn_gate[0], B, T, K.cout, st, nsc, nsh, npre, 0.3f, next_s32); }))) return rc;
            pre_is_s32 = next_s32;
        }
        std::swap(x, xn);
        h->rn_dbg_x = x; h->rn_dbg_T = T; h->rn_dbg_C = K.cout;
        if (stop_after == bi + 1) return SVHIP_OK;
        if (snap_at == bi + 1 && b0 == 0 && npre && (rc = snapshot(npre, T, K.cout))) return rc;
    }
    // aggregation: attentive statistics pooling                                          RawNet2_custom.py:215-224
    const int M = B * T;
    // (pre = lrelu(bn_before_agg(x)) came out of block 7's AFMS pass)
    if ((rc = conv_gemm(h, "rn_gemm", h->rn_att0, pre, 512, hb, 128, M, ACT_LRELU001))) return rc;
    float* rn_logits = h->rn_logits + (size_t)b0 * T * 512;
    if ((rc = conv_gemm(h, "rn_gemm", h->rn_att3, hb, 128, rn_logits, 512, M, ACT_NONE, ACT_NONE, nullptr, 0, nullptr, 0, true))) return rc;
    if ((rc = run(h, "rn_attn_pool", 0, [&]() { return launch_rn_attn_pool(rn_logits, pre, dt, B, T, 512, rn_pooled, st); }))) return rc;
    if ((rc = run(h, "rn_fc", 2.0 * B * h->rn_fc.N * h->rn_fc.K, [&]() {
             // (16-bit handles, full batches: the K-split MFMA form — fp32-grade handles keep ONE kernel for every batch size here)
             return launch_rowvec_linear(rn_pooled, 1024, h->rn_fc.W, h->rn_fc.bias, d_emb, c.embed_dim, B, c.embed_dim, 1024, ACT_NONE, st,
                                         h->bf16 && h->d_lin_part ? h->d_lin_part + (size_t)b0 * h->lin_part_per_utt : nullptr, true);
         }))) return rc;
    return SVHIP_OK;
}

// whole batch: one slice, or `lanes` slices on as many streams, so that the small and under-filled kernels of one slice (the late
// blocks are grids of 86 - 400 workgroups, the AFMS passes are latency-bound) run beside the big ones of another
int rawnet2_forward(svhip_handle* h, const float* d_wav, int B) {
    int rc = SVHIP_OK;
    const int lanes = (h->lanes > 1 && B >= 16 * h->lanes && h->opt.rn_stop < 0) ? h->lanes : 1;
    if (lanes > 1) {
        const int per = ((B + lanes - 1) / lanes + 3) & ~3;
        SV_HIP(h, hipEventRecord(h->lane_ev[4], h->stream));
        for (int l = 0; l < lanes && !rc; ++l) {
            const int b0 = l * per, n = std::min(per, B - b0);
            if (n <= 0) break;
            SV_HIP(h, hipStreamWaitEvent(h->lane_stream[l], h->lane_ev[4], 0));
            h->cur = h->lane_stream[l];
            rc = rawnet2_forward_part(h, d_wav, b0, n);
            h->cur = h->stream;
            if (rc) break;
            SV_HIP(h, hipEventRecord(h->lane_ev[l], h->lane_stream[l]));
            SV_HIP(h, hipStreamWaitEvent(h->stream, h->lane_ev[l], 0));
        }
    } else {
        h->cur = h->stream;
        rc = rawnet2_forward_part(h, d_wav, 0, B);
    }
    if (!rc) h->lastB = B;
    return rc;
}

int check_ready(svhip_handle* h, int B) {
    if (!h) return SVHIP_ERR_INVALID;
    if (!h->finalized) SV_FAIL(h, SVHIP_ERR_STATE, "weights not finalized (call svhip_finalize_weights first)");
    if (B <= 0 || B > h->cfg.max_batch) SV_FAIL(h, SVHIP_ERR_INVALID, "batch %d outside [1, max_batch=%d]", B, h->cfg.max_batch);
    return SVHIP_OK;
}

// The numeric status of the forwards since the last reset (the stream must be idle): the kernels that raise a bit also set the mapped
// host flag, so the common case costs one host load.  When both bits are set the RANGE report is the one returned: it names the cause (an input the
// planes cannot carry ends as inf / NaN embeddings) and carries the count of non-finite embedding values in its text.
enum { SVHIP_STATUS_NONFINITE = 1, SVHIP_STATUS_RANGE = 2 };
int numeric_status(svhip_handle* h, bool reset) {
    if (!h->host_flag || !*h->host_flag) return SVHIP_OK;
    uint32_t st[4] = {0, 0, 0, 0};
    SV_HIP(h, hipMemcpy(st, h->d_status, 16, hipMemcpyDeviceToHost));
    if (reset) {
        SV_HIP(h, hipMemset(h->d_status, 0, 16));
        *h->host_flag = 0;
    }
    // (the range report outranks the non-finite one: it names the cause — an input the planes cannot carry ends as inf / NaN embeddings)
    if (st[0] & SVHIP_STATUS_RANGE)
        SV_FAIL(h, SVHIP_ERR_RANGE, "%u input feature value(s) exceed 65504 in magnitude (or are not finite): SVHIP_F32X3 carries operands as IEEE-half "
                "hi | lo planes and cannot represent them; %u embedding value(s) came out non-finite (normalise the input - log_input / input_norm - "
                "or use compute = f32)", st[2], st[1]);
    if (st[0] & SVHIP_STATUS_NONFINITE)
        SV_FAIL(h, SVHIP_ERR_NONFINITE, "%u embedding value(s) are not finite%s (the embeddings were written as computed)", st[1],
                h->f16 ? ": an fp16 activation overflowed 65504 (or the input was not finite) - this checkpoint needs compute = f32 (exact); bf16 is "
                         "range-safe and fast but loses accuracy on RawNet2 (bf16 weight rounding)"
                : h->x3 ? ": a GEMM operand exceeded 65504, the range of SVHIP_F32X3's half-precision hi | lo planes (or the input was not finite) - "
                          "use compute = f32 (exact); bf16 is range-safe and fast at 16-bit accuracy"
                        : ": the input was not finite, or the weights overflow fp32");
    return SVHIP_OK;
}

int finish(svhip_handle* h, int flags) {
    if (flags & SVHIP_ASYNC) return SVHIP_OK;
    SV_HIP(h, hipStreamSynchronize(h->stream));
    return numeric_status(h, true);
}

// the embeddings leave the workspace: device output through emb_out_kernel (copy + finite check in one pass), host output checked in
// place and copied
int emit_embeddings(svhip_handle* h, int B, float* emb_out, int flags) {
    const int n = B * h->cfg.embed_dim;
    h->cur = h->stream;
    float* dst = (flags & SVHIP_OUT_DEVICE) ? emb_out : h->d_emb;
    int rc = run(h, "emb_out", 0, [&]() { return launch_emb_out(h->d_emb, dst, n, h->d_status, h->host_flag_dev, h->stream); });
    if (rc) return rc;
    if (!(flags & SVHIP_OUT_DEVICE)) SV_HIP(h, hipMemcpyAsync(emb_out, h->d_emb, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
    return SVHIP_OK;
}

}  // namespace

namespace svhip {
hipStream_t handle_stream(svhip_handle* h) { return h->stream; }
int handle_device(const svhip_handle* h) { return h->cfg.device; }
void handle_set_error(svhip_handle* h, const char* msg) { h->err = msg ? msg : ""; }
void*& handle_comm(svhip_handle* h) { return h->comm; }
int handle_run(svhip_handle* h, const char* label, const std::function<hipError_t()>& launch) {
    const std::string keep = h->err;             // a failing launch may have left a more specific message
    h->err.clear();
    h->cur = h->stream;
    const int rc = run(h, label, 0, [&]() { return launch(); });
    if (rc && !keep.empty() && h->err.empty()) h->err = keep;
    return rc;
}
}  // namespace svhip

// =====================================================================================================
extern "C" {

void svhip_default_config(svhip_config* c) {
    memset(c, 0, sizeof(*c));
    c->struct_size = (int32_t)sizeof(svhip_config);
    c->model = SVHIP_MODEL_ECAPA;
    c->compute = SVHIP_F32;
    c->device = 0;
    c->channels = 1024;
    c->n_mels = 80;
    c->embed_dim = 192;
    c->max_batch = 8;
    c->samples = 32000;
    c->log_input = 1;
    c->input_norm = 0;
    c->fb_sr = 8000; c->n_fft = 512; c->win_length = 200; c->hop_length = 80;
    c->fmin = 0.0f; c->fmax = -1.0f; c->preemph = 0.97f;
    c->stream = nullptr;
}

int svhip_abi_version(void) { return SVHIP_ABI_VERSION; }

const char* svhip_last_error(const svhip_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int svhip_create(const svhip_config* cfg, svhip_handle** out) {
    if (!cfg || !out || cfg->struct_size != (int32_t)sizeof(svhip_config)) { g_create_error = "bad config / struct_size"; return SVHIP_ERR_INVALID; }
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) { g_create_error = std::string("no HIP device: ") + hipGetErrorString(e); return SVHIP_ERR_HIP; }
    if (cfg->device < 0 || cfg->device >= ndev) { g_create_error = "device ordinal out of range"; return SVHIP_ERR_INVALID; }
    if (cfg->model != SVHIP_MODEL_ECAPA && cfg->model != SVHIP_MODEL_RAWNET2 && cfg->model != SVHIP_MODEL_NONE) { g_create_error = "unknown model"; return SVHIP_ERR_INVALID; }
    if (cfg->model == SVHIP_MODEL_ECAPA && (cfg->channels <= 0 || cfg->channels % 64 != 0)) { g_create_error = "ECAPA channels must be a positive multiple of 64"; return SVHIP_ERR_INVALID; }
    if (cfg->compute != SVHIP_F32 && cfg->compute != SVHIP_BF16 && cfg->compute != SVHIP_F32X3 && cfg->compute != SVHIP_F16) { g_create_error = "unknown compute mode"; return SVHIP_ERR_INVALID; }
    if (cfg->compute == SVHIP_F16 && cfg->model != SVHIP_MODEL_RAWNET2) { g_create_error = "SVHIP_F16 is RawNet2's 16-bit mode (ECAPA's is SVHIP_BF16)"; return SVHIP_ERR_UNSUPPORTED; }
    if (cfg->model == SVHIP_MODEL_RAWNET2 && cfg->samples < 251 + 3 * 3 * 3 * 3 * 3 * 3 * 3) { g_create_error = "RawNet2 needs at least 2438 samples"; return SVHIP_ERR_INVALID; }
    if (cfg->n_mels <= 0 || cfg->n_mels % 8 != 0 || cfg->max_batch <= 0 || cfg->samples < cfg->n_fft || cfg->hop_length <= 0) { g_create_error = "bad n_mels / max_batch / samples"; return SVHIP_ERR_INVALID; }
    if ((e = hipSetDevice(cfg->device)) != hipSuccess) { g_create_error = hipGetErrorString(e); return SVHIP_ERR_HIP; }
    svhip_handle* h = new svhip_handle();
    h->cfg = *cfg;
    h->f16 = cfg->compute == SVHIP_F16;
    h->bf16 = cfg->compute == SVHIP_BF16 || h->f16;
    h->dt = h->f16 ? DT_F16 : h->bf16 ? DT_BF16 : DT_F32;
    h->x3 = cfg->compute == SVHIP_F32X3;
    {   // the developer switches' defaults come from the environment, once
        auto flag = [](const char* n) { return getenv(n) != nullptr ? 1 : 0; };
        auto is1 = [](const char* n) { const char* e = getenv(n); return (e && e[0] == '1') ? 1 : 0; };
        auto num = [](const char* n, int dflt) { const char* e = getenv(n); return e ? atoi(e) : dflt; };
        svhip_handle::DevOpts& o = h->opt;
        o.layer_labels = flag("SVHIP_LAYER_LABELS"); o.x3_keep_f32 = flag("SVHIP_X3_KEEP_F32"); o.r2_big = is1("SVHIP_R2_BIG");
        o.asp_v1 = is1("SVHIP_ASP_V1"); o.rn_stop = num("SVHIP_RN_STOP", -1); o.rn_snap = num("SVHIP_RN_SNAP", -1);
        o.rn_unfused = flag("SVHIP_RN_UNFUSED"); o.asnorm_slab = flag("SVHIP_ASNORM_SLAB"); o.asnorm_f32mfma = flag("SVHIP_ASNORM_F32MFMA"); o.asnorm_x6 = flag("SVHIP_ASNORM_X6"); o.asnorm_w32 = flag("SVHIP_ASNORM_W32"); o.score_tiled = flag("SVHIP_SCORE_TILED"); o.score_f32mfma = flag("SVHIP_SCORE_F32MFMA");
        o.fbank32 = is1("SVHIP_FBANK32"); o.rn_sinc_full = is1("SVHIP_RN_SINC_FULL"); o.fbank_unfused = is1("SVHIP_FBANK_UNFUSED"); o.pw3_cus = num("SVHIP_PW3_CUS", -1); o.pw3_tail_off = is1("SVHIP_PW3_TAIL_OFF"); o.pw4 = is1("SVHIP_PW4"); o.asnorm_2s = is1("SVHIP_ASNORM_2S"); o.cv_off = is1("SVHIP_CV_OFF"); o.n128_off = is1("SVHIP_N128_OFF"); o.r2_slices = num("SVHIP_R2_SLICES", -1); o.rn_tail_big = is1("SVHIP_RN_TAIL_BIG"); o.rn_sinc_f32 = is1("SVHIP_RN_SINC_F32"); o.rn_step_off = is1("SVHIP_RN_STEP_OFF"); o.rn_pool_off = is1("SVHIP_RN_POOL_OFF");
    }
    h->esz = h->bf16 ? 2 : 4;
    h->T = cfg->samples / cfg->hop_length + 1;
    if (cfg->stream) { h->stream = reinterpret_cast<hipStream_t>(cfg->stream); h->own_stream = false; }
    else {
        if ((e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess) { g_create_error = hipGetErrorString(e); delete h; return SVHIP_ERR_HIP; }
        h->own_stream = true;
    }
    h->cur = h->stream;
    {
        int ncu = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, cfg->device) == hipSuccess && ncu > 0) h->num_cu = ncu;
    }
    {
        // batch slices on separate streams, SVHIP_LANES = 1 .. 4 (default 1).  Measured at B = 256: ECAPA -1.5 .. +3.7 % with two
        // (its big GEMMs fill the chip either way); RawNet2 +3 % with two or three, -22 % with four — slices of ONE call all sit in
        // the same phase of the network, so little complements.  What does pay for RawNet2 is whole batches in flight on
        // separate handles / streams (+12 .. 21 %, bench.py `rawnet2_3_streams`): a serving-loop choice, not a library default.
        const char* le = getenv("SVHIP_LANES");
        h->lanes = le ? atoi(le) : 1;
        if (h->lanes < 1 || h->lanes > 4 || cfg->model == SVHIP_MODEL_NONE) h->lanes = 1;
        if (cfg->model == SVHIP_MODEL_ECAPA && h->lanes > 2) h->lanes = 2;
        if (h->lanes > 1) {
            for (int i = 0; i < h->lanes; ++i) (void)hipStreamCreateWithFlags(&h->lane_stream[i], hipStreamNonBlocking);
            for (int i = 0; i < 5; ++i) (void)hipEventCreateWithFlags(&h->lane_ev[i], hipEventDisableTiming);
        }
    }
    int rc = build_fbank_tables(h);
    h->fb.force32 = h->opt.fbank32;
    if (rc == SVHIP_OK) rc = alloc_workspace(h);
    if (rc != SVHIP_OK) { g_create_error = h->err; svhip_destroy(h); return rc; }
    if (cfg->model == SVHIP_MODEL_NONE) h->finalized = true;
    *out = h;
    return SVHIP_OK;
}

int svhip_destroy(svhip_handle* h) {
    if (!h) return SVHIP_OK;
    (void)hipSetDevice(h->cfg.device);
    if (h->comm) (void)svhip_comm_destroy(h);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->crop_pcm) (void)hipFree(h->crop_pcm);
    if (h->aux_stream) { (void)hipStreamSynchronize(h->aux_stream); (void)hipStreamDestroy(h->aux_stream); }
    for (hipEvent_t e : h->aux_ev) if (e) (void)hipEventDestroy(e);
    for (void* q : h->scr) if (q) (void)hipFree(q);
    for (auto& sl : h->crop_slot) {
        if (sl.host) (void)hipHostFree(sl.host);
        if (sl.dev) (void)hipFree(sl.dev);
        if (sl.done) (void)hipEventDestroy(sl.done);
    }
    for (void* p : h->allocs) (void)hipFree(p);
    if (h->host_flag) (void)hipHostFree(h->host_flag);
    prof_collect(h);
    for (hipEvent_t e : h->ev_free) (void)hipEventDestroy(e);
    for (int i = 0; i < 4; ++i) if (h->lane_stream[i]) { (void)hipStreamSynchronize(h->lane_stream[i]); (void)hipStreamDestroy(h->lane_stream[i]); }
    for (int i = 0; i < 5; ++i) if (h->lane_ev[i]) (void)hipEventDestroy(h->lane_ev[i]);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return SVHIP_OK;
}

int svhip_synchronize(svhip_handle* h) {
    if (!h) return SVHIP_ERR_INVALID;
    SV_HIP(h, hipStreamSynchronize(h->stream));
    return numeric_status(h, true);
}

int svhip_numeric_status(svhip_handle* h, int32_t reset) {
    if (!h) return SVHIP_ERR_INVALID;
    SV_HIP(h, hipSetDevice(h->cfg.device));
    SV_HIP(h, hipStreamSynchronize(h->stream));
    return numeric_status(h, reset != 0);
}

int svhip_load_tensor(svhip_handle* h, const char* name, const void* data, const int64_t* shape, int32_t ndim, int32_t dtype) {
    if (!h || !name || !shape || ndim < 0 || ndim > 4) return SVHIP_ERR_INVALID;
    if (!data) SV_FAIL(h, SVHIP_ERR_INVALID, "null data for %s (a 0-d tensor still holds one element)", name);
    if (h->finalized) SV_FAIL(h, SVHIP_ERR_STATE, "weights already finalized");
    std::map<std::string, std::vector<int64_t>> spec;
    model_spec(h->cfg, spec);
    auto it = spec.find(name);
    if (it == spec.end()) SV_FAIL(h, SVHIP_ERR_INVALID, "%s is not in the model.", name);
    std::vector<int64_t> shp(shape, shape + ndim);
    if (shp != it->second) SV_FAIL(h, SVHIP_ERR_INVALID, "Wrong parameter shape: %s", name);
    HostTensor t;
    t.shape = shp;
    const int64_t n = t.numel();
    t.data.resize((size_t)n);
    if (dtype == SVHIP_F32) memcpy(t.data.data(), data, (size_t)n * 4);
    else if (dtype == SVHIP_I64) for (int64_t i = 0; i < n; ++i) t.data[i] = (float)reinterpret_cast<const int64_t*>(data)[i];
    else SV_FAIL(h, SVHIP_ERR_INVALID, "unsupported dtype %d for %s", dtype, name);
    h->host_w[name] = std::move(t);
    return SVHIP_OK;
}

int svhip_finalize_weights(svhip_handle* h) {
    if (!h) return SVHIP_ERR_INVALID;
    if (h->finalized) SV_FAIL(h, SVHIP_ERR_STATE, "weights already finalized");
    SV_HIP(h, hipSetDevice(h->cfg.device));
    std::map<std::string, std::vector<int64_t>> spec;
    model_spec(h->cfg, spec);
    for (auto& kv : spec)
        if (!h->host_w.count(kv.first) && kv.first.find("num_batches_tracked") == std::string::npos)
            SV_FAIL(h, SVHIP_ERR_MISSING, "tensor %s was never loaded", kv.first.c_str());
    int rc = SVHIP_ERR_UNSUPPORTED;
    if (h->cfg.model == SVHIP_MODEL_ECAPA) rc = finalize_ecapa(h);
    else if (h->cfg.model == SVHIP_MODEL_RAWNET2) rc = finalize_rawnet2(h);
    else SV_FAIL(h, SVHIP_ERR_UNSUPPORTED, "model %d has no forward path in this build", h->cfg.model);
    if (rc) return rc;
    SV_HIP(h, hipDeviceSynchronize());
    h->host_w.clear();
    h->finalized = true;
    return SVHIP_OK;
}

int svhip_load_blob(svhip_handle* h, const char* path) {
    if (!h || !path) return SVHIP_ERR_INVALID;
    if (h->finalized) SV_FAIL(h, SVHIP_ERR_STATE, "weights already finalized");
    svhip_blob* b = nullptr;
    if (int rc = svhip_blob_open(path, &b)) SV_FAIL(h, rc, "%s", svhip_blob_last_error());
    if (svhip_blob_model(b) != h->cfg.model) {
        const int m = svhip_blob_model(b);
        svhip_blob_close(b);
        SV_FAIL(h, SVHIP_ERR_INVALID, "%s holds weights of model %d, this handle is model %d", path, m, h->cfg.model);
    }
    std::map<std::string, std::vector<int64_t>> spec;
    model_spec(h->cfg, spec);
    const int32_t n = svhip_blob_count(b);
    for (int32_t i = 0; i < n; ++i) {
        const char* name; const void* data; int64_t shape[4]; int32_t ndim, dtype;
        svhip_blob_tensor(b, i, &name, &data, shape, &ndim, &dtype);
        if (!spec.count(name)) continue;                       // e.g. loss-head tensors of a training checkpoint
        if (int rc = svhip_load_tensor(h, name, data, shape, ndim, dtype)) { svhip_blob_close(b); return rc; }
    }
    svhip_blob_close(b);
    return svhip_finalize_weights(h);
}

int svhip_fbank(svhip_handle* h, const float* wav, int32_t B, int32_t L, float* mel_out, int32_t flags) {
    if (!h || !wav || !mel_out) return SVHIP_ERR_INVALID;
    if (B <= 0 || B > h->cfg.max_batch) SV_FAIL(h, SVHIP_ERR_INVALID, "batch %d outside [1, max_batch=%d]", B, h->cfg.max_batch);
    if (L != h->cfg.samples) SV_FAIL(h, SVHIP_ERR_INVALID, "L=%d but the handle was created for %d samples", L, h->cfg.samples);
    if ((flags & SVHIP_ASYNC) && (flags & (SVHIP_IN_DEVICE | SVHIP_OUT_DEVICE)) != (SVHIP_IN_DEVICE | SVHIP_OUT_DEVICE))
        SV_FAIL(h, SVHIP_ERR_INVALID, "SVHIP_ASYNC needs device pointers");
    SV_HIP(h, hipSetDevice(h->cfg.device));
    const float* d_in = wav;
    if (!(flags & SVHIP_IN_DEVICE)) {
        SV_HIP(h, hipMemcpyAsync(h->d_wav, wav, (size_t)B * L * 4, hipMemcpyHostToDevice, h->stream));
        d_in = h->d_wav;
    }
    float* d_out = (flags & SVHIP_OUT_DEVICE) ? mel_out : h->d_feat;
    const int T = h->T;
    int rc = run(h, "fbank", 0, [&]() { return launch_fbank(h->fb, d_in, B, L, T, d_out, h->stream); });
    if (rc) return rc;
    if (!(flags & SVHIP_OUT_DEVICE))
        SV_HIP(h, hipMemcpyAsync(mel_out, d_out, (size_t)B * h->cfg.n_mels * T * 4, hipMemcpyDeviceToHost, h->stream));
    return finish(h, flags);
}

int svhip_embed_features(svhip_handle* h, const float* feat, int32_t B, int32_t T, float* emb_out, int32_t flags) {
    int rc = check_ready(h, B);
    if (rc) return rc;
    if (!feat || !emb_out) SV_FAIL(h, SVHIP_ERR_INVALID, "null pointer");
    if (h->cfg.model != SVHIP_MODEL_ECAPA) SV_FAIL(h, SVHIP_ERR_UNSUPPORTED, "embed_features needs a spectral model (ECAPA)");
    if (T != h->T) SV_FAIL(h, SVHIP_ERR_INVALID, "T=%d but the handle was created for T=%d frames", T, h->T);
    if ((flags & SVHIP_ASYNC) && (flags & (SVHIP_IN_DEVICE | SVHIP_OUT_DEVICE)) != (SVHIP_IN_DEVICE | SVHIP_OUT_DEVICE))
        SV_FAIL(h, SVHIP_ERR_INVALID, "SVHIP_ASYNC needs device pointers");
    SV_HIP(h, hipSetDevice(h->cfg.device));
    const float* d_in = feat;
    if (!(flags & SVHIP_IN_DEVICE)) {
        SV_HIP(h, hipMemcpyAsync(h->d_feat, feat, (size_t)B * h->cfg.n_mels * T * 4, hipMemcpyHostToDevice, h->stream));
        d_in = h->d_feat;
    }
    h->feat_is_stale = false;
    if ((rc = ecapa_forward(h, d_in, B))) return rc;
    if ((rc = emit_embeddings(h, B, emb_out, flags))) return rc;
    return finish(h, flags);
}

int svhip_embed_wave(svhip_handle* h, const float* wav, int32_t B, int32_t L, float* emb_out, int32_t flags) {
    int rc = check_ready(h, B);
    if (rc) return rc;
    if (!wav || !emb_out) SV_FAIL(h, SVHIP_ERR_INVALID, "null pointer");
    if (L != h->cfg.samples) SV_FAIL(h, SVHIP_ERR_INVALID, "L=%d but the handle was created for %d samples", L, h->cfg.samples);
    if ((flags & SVHIP_ASYNC) && (flags & (SVHIP_IN_DEVICE | SVHIP_OUT_DEVICE)) != (SVHIP_IN_DEVICE | SVHIP_OUT_DEVICE))
        SV_FAIL(h, SVHIP_ERR_INVALID, "SVHIP_ASYNC needs device pointers");
    if (h->cfg.model != SVHIP_MODEL_ECAPA && h->cfg.model != SVHIP_MODEL_RAWNET2)
        SV_FAIL(h, SVHIP_ERR_UNSUPPORTED, "model %d has no forward path in this build", h->cfg.model);
    SV_HIP(h, hipSetDevice(h->cfg.device));
    const float* d_in = wav;
    if (!(flags & SVHIP_IN_DEVICE)) {
        SV_HIP(h, hipMemcpyAsync(h->d_wav, wav, (size_t)B * L * 4, hipMemcpyHostToDevice, h->stream));
        d_in = h->d_wav;
    }
    if (h->cfg.model == SVHIP_MODEL_RAWNET2) {
        if ((rc = rawnet2_forward(h, d_in, B))) return rc;
    } else {
        const int T = h->T;
        // bf16 handles without the instance-norm prologue: waveform -> the 16-bit operand of blocks.0 in two launches (fbank.hip, round 6)
        const bool fused = h->bf16 && !h->in_w && h->d_logmel && !h->opt.fbank_unfused && !h->opt.fbank32 && fbank_fused_supported(h->fb, L);
        if (fused) {
            if ((rc = run(h, "fbank_fused", 0, [&]() {
                     return launch_fbank_fused(h->fb, d_in, B, L, T, h->cfg.log_input, h->d_logmel, h->d_fpart, h->X_in, h->stream);
                 }))) return rc;
        } else if ((rc = run(h, "fbank", 0, [&]() { return launch_fbank(h->fb, d_in, B, L, T, h->d_feat, h->stream); }))) return rc;
        h->xin_ready = fused;
        h->feat_is_stale = fused;
        rc = ecapa_forward(h, h->d_feat, B);
        h->xin_ready = false;
        if (rc) return rc;
    }
    if ((rc = emit_embeddings(h, B, emb_out, flags))) return rc;
    return finish(h, flags);
}

int svhip_crop_pcm16(svhip_handle* h, const int16_t* pcm, int64_t n_samples, const int64_t* offsets, const int32_t* lengths,
                     int32_t n_files, int32_t num_eval, int32_t L, float* crops_out, int32_t flags) {
    if (!h || !pcm || !offsets || !lengths || !crops_out || n_files <= 0 || num_eval <= 0 || L <= 0 || n_samples <= 0) return SVHIP_ERR_INVALID;
    SV_HIP(h, hipSetDevice(h->cfg.device));
    const bool din = flags & SVHIP_IN_DEVICE, dout = flags & SVHIP_OUT_DEVICE;
    if ((flags & SVHIP_ASYNC) && !dout) SV_FAIL(h, SVHIP_ERR_INVALID, "SVHIP_ASYNC needs a device output pointer");
    if (!din)      // host metadata is range-checked before it reaches the GPU
        for (int f = 0; f < n_files; ++f)
            if (lengths[f] <= 0 || offsets[f] < 0 || offsets[f] + lengths[f] > n_samples)
                SV_FAIL(h, SVHIP_ERR_INVALID, "file %d: offset/length outside the PCM buffer", f);
    const void *d_pcm = pcm, *d_off = offsets, *d_len = lengths;
    void* d_out = crops_out;
    const size_t out_bytes = (size_t)n_files * num_eval * L * 4;
    svhip_handle::CropSlot* slot = nullptr;
    if (!din) {
        // PCM: one device buffer (the copy is ordered behind the previous call's kernel on this stream); pageable host memory makes
        // the copy call itself block until the bytes are staged, pinned memory makes it truly asynchronous
        const size_t pcm_bytes = (size_t)n_samples * 2;
        if (h->crop_pcm_cap < pcm_bytes) {
            SV_HIP(h, hipStreamSynchronize(h->stream));
            if (h->crop_pcm) (void)hipFree(h->crop_pcm);
            h->crop_pcm = nullptr; h->crop_pcm_cap = 0;
            SV_HIP(h, hipMalloc(&h->crop_pcm, pcm_bytes + pcm_bytes / 4));
            h->crop_pcm_cap = pcm_bytes + pcm_bytes / 4;
        }
        // metadata: copied into a pinned slot of the handle, so the caller's arrays are free on return
        slot = &h->crop_slot[h->crop_next];
        h->crop_next = (h->crop_next + 1) & 3;
        if (slot->busy) { SV_HIP(h, hipEventSynchronize(slot->done)); slot->busy = false; }
        const size_t meta = (size_t)n_files * 12;
        if (slot->cap < meta) {
            if (slot->host) (void)hipHostFree(slot->host);
            if (slot->dev) (void)hipFree(slot->dev);
            slot->host = slot->dev = nullptr; slot->cap = 0;
            SV_HIP(h, hipHostMalloc((void**)&slot->host, meta * 2, hipHostMallocDefault));
            SV_HIP(h, hipMalloc((void**)&slot->dev, meta * 2));
            slot->cap = meta * 2;
            if (!slot->done) SV_HIP(h, hipEventCreateWithFlags(&slot->done, hipEventDisableTiming));
        }
        memcpy(slot->host, offsets, (size_t)n_files * 8);
        memcpy(slot->host + (size_t)n_files * 8, lengths, (size_t)n_files * 4);
        SV_HIP(h, hipMemcpyAsync(h->crop_pcm, pcm, pcm_bytes, hipMemcpyHostToDevice, h->stream));
        SV_HIP(h, hipMemcpyAsync(slot->dev, slot->host, meta, hipMemcpyHostToDevice, h->stream));
        d_pcm = h->crop_pcm; d_off = slot->dev; d_len = slot->dev + (size_t)n_files * 8;
    }
    void* tmp_out = nullptr;
    if (!dout) { SV_HIP(h, hipMalloc(&tmp_out, out_bytes)); d_out = tmp_out; }
    h->cur = h->stream;
    int rc = run(h, "crop_pcm16", 0, [&]() { return launch_crop_pcm16((const int16_t*)d_pcm, (const int64_t*)d_off, (const int32_t*)d_len, n_files, num_eval, L, (float*)d_out, h->stream); });
    hipError_t e = hipSuccess;
    if (slot && !rc) { e = hipEventRecord(slot->done, h->stream); slot->busy = e == hipSuccess; }
    if (!rc && !dout && e == hipSuccess) e = hipMemcpyAsync(crops_out, d_out, out_bytes, hipMemcpyDeviceToHost, h->stream);
    if (!(flags & SVHIP_ASYNC)) { const hipError_t e2 = hipStreamSynchronize(h->stream); if (e == hipSuccess) e = e2; }
    if (tmp_out) (void)hipFree(tmp_out);
    if (rc) return rc;
    if (e != hipSuccess) SV_FAIL(h, SVHIP_ERR_HIP, "crop staging / copy failed: %s", hipGetErrorString(e));
    return SVHIP_OK;
}

int svhip_synth_waveforms(svhip_handle* h, uint64_t seed, int64_t first_utt, int32_t B, int32_t L, float* wav_out, int32_t flags) {
    if (!h || !wav_out || B <= 0 || L <= 0 || first_utt < 0) return SVHIP_ERR_INVALID;
    if (L % 4 != 0) SV_FAIL(h, SVHIP_ERR_INVALID, "L=%d must be a multiple of 4", L);
    if ((flags & SVHIP_ASYNC) && !(flags & SVHIP_OUT_DEVICE)) SV_FAIL(h, SVHIP_ERR_INVALID, "SVHIP_ASYNC needs device pointers");
    SV_HIP(h, hipSetDevice(h->cfg.device));
    float* d_out = wav_out;
    void* tmp = nullptr;
    if (!(flags & SVHIP_OUT_DEVICE)) {
        SV_HIP(h, hipMalloc(&tmp, (size_t)B * L * 4));
        d_out = reinterpret_cast<float*>(tmp);
    }
    h->cur = h->stream;
    int rc = run(h, "synth_wave", 0, [&]() { return launch_synth_wave(d_out, seed, first_utt, B, L, h->stream); });
    hipError_t e = hipSuccess;
    if (!rc && tmp) e = hipMemcpyAsync(wav_out, d_out, (size_t)B * L * 4, hipMemcpyDeviceToHost, h->stream);
    if (tmp || !(flags & SVHIP_ASYNC)) { const hipError_t e2 = hipStreamSynchronize(h->stream); if (e == hipSuccess) e = e2; }
    if (tmp) (void)hipFree(tmp);
    if (rc) return rc;
    SV_HIP(h, e);
    return SVHIP_OK;
}

// ---- scoring ------------------------------------------------------------------------------------------
namespace {
// a scratch slot of at least `bytes` (grown by half again; the old block is freed only after the stream has drained)
int scratch(svhip_handle* h, int slot, size_t bytes, void** out) {
    if (h->scr_cap[slot] < bytes || !h->scr[slot]) {
        if (h->scr[slot]) { SV_HIP(h, hipStreamSynchronize(h->stream)); if (h->aux_stream) SV_HIP(h, hipStreamSynchronize(h->aux_stream)); (void)hipFree(h->scr[slot]); h->scr[slot] = nullptr; h->scr_cap[slot] = 0; }
        const size_t cap = std::max<size_t>(bytes + bytes / 2, 256);
        if (hipMalloc(&h->scr[slot], cap) != hipSuccess) {
            if (hipMalloc(&h->scr[slot], std::max<size_t>(bytes, 256)) != hipSuccess) { h->scr[slot] = nullptr; SV_FAIL(h, SVHIP_ERR_NOMEM, "scratch slot %d: %zu bytes", slot, bytes); }
            h->scr_cap[slot] = std::max<size_t>(bytes, 256);
        } else {
            h->scr_cap[slot] = cap;
        }
    }
    *out = h->scr[slot];
    return SVHIP_OK;
}
// Scratch retention policy: the slab of the AS-norm slab path (up to 2 GiB) and the staging copies of HOST-pointer calls
// (N x D x 4 bytes for the embedding matrix) are released at the end of the call once they exceed 256 MiB — they would otherwise sit
// beside the model engines' workspaces for the life of the process-wide scoring handle.  The slots of the device-resident fast path
// (candidate lists, cohort planes: re-used every call) stay; svhip_trim_scratch frees everything.
void release_big_scratch(svhip_handle* h, bool staged_host) {
    const size_t cap = (size_t)256 << 20;
    auto drop = [&](int slot) {
        if (h->scr[slot] && h->scr_cap[slot] > cap) {
            (void)hipStreamSynchronize(h->stream);
            if (h->aux_stream) (void)hipStreamSynchronize(h->aux_stream);
            (void)hipFree(h->scr[slot]);
            h->scr[slot] = nullptr; h->scr_cap[slot] = 0;
        }
    };
    drop(svhip_handle::SCR_SLAB);
    if (staged_host)
        for (int s_ : {svhip_handle::SCR_IN0, svhip_handle::SCR_IN1, svhip_handle::SCR_IN2, svhip_handle::SCR_IN3, svhip_handle::SCR_IN4,
                       svhip_handle::SCR_OUT0, svhip_handle::SCR_OUT1, svhip_handle::SCR_OUT2, svhip_handle::SCR_SPLIT}) drop(s_);
}
struct TempBuf {      // device staging for host-pointer calls, in a scratch slot of the handle
    svhip_handle* h; int slot;
    int in(const void* src, size_t bytes, bool is_dev, const void** out) {
        if (is_dev) { *out = src; return SVHIP_OK; }
        void* d;
        if (int rc = scratch(h, slot, bytes, &d)) return rc;
        SV_HIP(h, hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, h->stream));
        *out = d;
        return SVHIP_OK;
    }
    int out(void* dst, size_t bytes, bool is_dev, void** o) {
        if (is_dev) { *o = dst; return SVHIP_OK; }
        return scratch(h, slot, bytes, o);
    }
};
}  // namespace

int svhip_l2norm(svhip_handle* h, float* E, int64_t N, int32_t D, int32_t flags) {
    if (!h || !E || N < 0 || D <= 0) return SVHIP_ERR_INVALID;
    SV_HIP(h, hipSetDevice(h->cfg.device));
    const bool dev = flags & SVHIP_IN_DEVICE;
    TempBuf t{h, svhip_handle::SCR_IN0};
    const void* d;
    int rc = t.in(E, (size_t)N * D * 4, dev, &d);
    if (rc) return rc;
    float* de = const_cast<float*>(reinterpret_cast<const float*>(d));
    if ((rc = run(h, "l2norm", 0, [&]() { return launch_l2norm(de, N, D, h->stream); }))) return rc;
    if (!dev) SV_HIP(h, hipMemcpyAsync(E, de, (size_t)N * D * 4, hipMemcpyDeviceToHost, h->stream));
    if (!dev || !(flags & SVHIP_ASYNC)) SV_HIP(h, hipStreamSynchronize(h->stream));
    return SVHIP_OK;
}

static int pairs_common(svhip_handle* h, int mode, const float* E, int64_t N, int32_t D, const float* mu, const float* sigma,
                        const int32_t* ia, const int32_t* ib, int64_t P, float* out, int32_t flags) {
    if (!h || !E || !ia || !ib || !out || N <= 0 || D <= 0 || P < 0) return SVHIP_ERR_INVALID;
    if (mode == 1 && (!mu || !sigma)) return SVHIP_ERR_INVALID;
    SV_HIP(h, hipSetDevice(h->cfg.device));
    const bool din = flags & SVHIP_IN_DEVICE, dout = flags & SVHIP_OUT_DEVICE;
    if (!din) {   // host indices are range-checked before they reach the GPU
        for (int64_t p = 0; p < P; ++p)
            if (ia[p] < 0 || ia[p] >= N || ib[p] < 0 || ib[p] >= N) SV_FAIL(h, SVHIP_ERR_INVALID, "pair %lld indexes outside [0, %lld)", (long long)p, (long long)N);
    }
    TempBuf tE{h, svhip_handle::SCR_IN0}, tA{h, svhip_handle::SCR_IN1}, tB{h, svhip_handle::SCR_IN2}, tM{h, svhip_handle::SCR_IN3},
        tS{h, svhip_handle::SCR_IN4}, tO{h, svhip_handle::SCR_OUT0};
    const void *dE, *dA, *dB, *dM = nullptr, *dS = nullptr;
    void* dO;
    int rc;
    if ((rc = tE.in(E, (size_t)N * D * 4, din, &dE))) return rc;
    if ((rc = tA.in(ia, (size_t)P * 4, din, &dA))) return rc;
    if ((rc = tB.in(ib, (size_t)P * 4, din, &dB))) return rc;
    if (mode == 1) {
        if ((rc = tM.in(mu, (size_t)N * 4, din, &dM))) return rc;
        if ((rc = tS.in(sigma, (size_t)N * 4, din, &dS))) return rc;
    }
    if ((rc = tO.out(out, (size_t)P * 4, dout, &dO))) return rc;
    if (mode == 0)
        rc = run(h, "score_pairs", 2.0 * P * D, [&]() { return launch_score_pairs((const float*)dE, D, (const int32_t*)dA, (const int32_t*)dB, P, (float*)dO, h->stream); });
    else
        rc = run(h, "asnorm_pairs", 2.0 * P * D, [&]() { return launch_asnorm_pairs((const float*)dE, D, (const float*)dM, (const float*)dS, (const int32_t*)dA, (const int32_t*)dB, P, (float*)dO, h->stream); });
    if (rc) return rc;
    if (!dout) SV_HIP(h, hipMemcpyAsync(out, dO, (size_t)P * 4, hipMemcpyDeviceToHost, h->stream));
    if (!(din && dout && (flags & SVHIP_ASYNC))) SV_HIP(h, hipStreamSynchronize(h->stream));
    return SVHIP_OK;
}

int svhip_score_pairs(svhip_handle* h, const float* E, int64_t N, int32_t D, const int32_t* ia, const int32_t* ib, int64_t P,
                      float* out, int32_t flags) {
    return pairs_common(h, 0, E, N, D, nullptr, nullptr, ia, ib, P, out, flags);
}

int svhip_asnorm_pairs(svhip_handle* h, const float* E, int64_t N, int32_t D, const float* mu, const float* sigma,
                       const int32_t* ia, const int32_t* ib, int64_t P, float* out, int32_t flags) {
    return pairs_common(h, 1, E, N, D, mu, sigma, ia, ib, P, out, flags);
}

static int score_trials_impl(svhip_handle* h, int32_t mode, float pexp, const float* F, int64_t n_files, int32_t n_crops, int32_t D, const int32_t* ia,
                       const int32_t* ib, int64_t P, float* out, int32_t flags) {
    if (!h || !F || !ia || !ib || !out || n_files <= 0 || n_crops <= 0 || D <= 0 || P < 0) return SVHIP_ERR_INVALID;
    if (mode < SVHIP_TRIAL_COSINE || mode > SVHIP_TRIAL_PDIST) SV_FAIL(h, SVHIP_ERR_INVALID, "unknown trial scoring mode %d", mode);
    SV_HIP(h, hipSetDevice(h->cfg.device));
    const bool din = flags & SVHIP_IN_DEVICE, dout = flags & SVHIP_OUT_DEVICE;
    if (!din)
        for (int64_t p = 0; p < P; ++p)
            if (ia[p] < 0 || ia[p] >= n_files || ib[p] < 0 || ib[p] >= n_files) SV_FAIL(h, SVHIP_ERR_INVALID, "trial %lld indexes outside [0, %lld)", (long long)p, (long long)n_files);
    TempBuf tF{h, svhip_handle::SCR_IN0}, tA{h, svhip_handle::SCR_IN1}, tB{h, svhip_handle::SCR_IN2}, tO{h, svhip_handle::SCR_OUT0};
    const void *dF, *dA, *dB;
    void* dO;
    int rc;
    if ((rc = tF.in(F, (size_t)n_files * n_crops * D * 4, din, &dF))) return rc;
    if ((rc = tA.in(ia, (size_t)P * 4, din, &dA))) return rc;
    if ((rc = tB.in(ib, (size_t)P * 4, din, &dB))) return rc;
    if ((rc = tO.out(out, (size_t)P * 4, dout, &dO))) return rc;
    if ((rc = run(h, "score_trials", 2.0 * P * n_crops * D, [&]() {
             return launch_trial_crops(mode, pexp, (const float*)dF, n_crops, D, (const int32_t*)dA, (const int32_t*)dB, P, (float*)dO, h->stream);
         }))) return rc;
    if (!dout) SV_HIP(h, hipMemcpyAsync(out, dO, (size_t)P * 4, hipMemcpyDeviceToHost, h->stream));
    if (!(din && dout && (flags & SVHIP_ASYNC))) SV_HIP(h, hipStreamSynchronize(h->stream));
    return SVHIP_OK;
}

int svhip_score_trials(svhip_handle* h, int32_t mode, const float* F, int64_t n_files, int32_t n_crops, int32_t D, const int32_t* ia,
                       const int32_t* ib, int64_t P, float* out, int32_t flags) {
    return score_trials_impl(h, mode, 2.0f, F, n_files, n_crops, D, ia, ib, P, out, flags);
}

int svhip_score_trials_pnorm(svhip_handle* h, float p, const float* F, int64_t n_files, int32_t n_crops, int32_t D, const int32_t* ia,
                             const int32_t* ib, int64_t P, float* out, int32_t flags) {
    if (h && p != p) SV_FAIL(h, SVHIP_ERR_INVALID, "pnorm: p is NaN");
    return score_trials_impl(h, SVHIP_TRIAL_PNORM, p, F, n_files, n_crops, D, ia, ib, P, out, flags);
}

int svhip_mean_crops(svhip_handle* h, const float* F, int64_t n_files, int32_t n_crops, int32_t D, float* out, int32_t flags) {
    if (!h || !F || !out || n_files <= 0 || n_crops <= 0 || D <= 0) return SVHIP_ERR_INVALID;
    SV_HIP(h, hipSetDevice(h->cfg.device));
    const bool din = flags & SVHIP_IN_DEVICE, dout = flags & SVHIP_OUT_DEVICE;
    TempBuf tF{h, svhip_handle::SCR_IN0}, tO{h, svhip_handle::SCR_OUT0};
    const void* dF;
    void* dO;
    int rc;
    if ((rc = tF.in(F, (size_t)n_files * n_crops * D * 4, din, &dF))) return rc;
    if ((rc = tO.out(out, (size_t)n_files * D * 4, dout, &dO))) return rc;
    if ((rc = run(h, "mean_crops", 0, [&]() { return launch_mean_crops((const float*)dF, n_files, n_crops, D, (float*)dO, h->stream); }))) return rc;
    if (!dout) SV_HIP(h, hipMemcpyAsync(out, dO, (size_t)n_files * D * 4, hipMemcpyDeviceToHost, h->stream));
    if (!(din && dout && (flags & SVHIP_ASYNC))) SV_HIP(h, hipStreamSynchronize(h->stream));
    return SVHIP_OK;
}

// out (Na, Nb) = A @ B^T.  The route is decided ONCE (split_b and score_gemm used to re-derive it from different predicates):
//   SCORE_H3W     rows of A in registers, B streamed past them as half planes (asnorm_fused.hip: score_h3w; D = 192 / 256, aligned operands);
//                 the kernel's launcher fills the planes itself (and scales both operands by exact powers of two)
//   SCORE_WORDS   the tiled GEMM with B as (hi half << 16 | lo half) words, A split in registers (gemm_pw's x3 form)
//   SCORE_F32MFMA the exact fp32 MFMA GEMM (option score_f32mfma, or shapes gemm_pw's split form does not take)
// The split forms run as three fp16 MFMAs per product on every handle since round 4 (a score of unit vectors within ~4e-8 of the float64
// oracle, the exact fp32 MFMA 3.5e-8) at twice the fp32 matrix rate.
enum ScoreRoute { SCORE_F32MFMA = 0, SCORE_WORDS = 1, SCORE_H3W = 2 };
static ScoreRoute score_route(const svhip_handle* h, const float* dA, int64_t Na, const float* dB, int64_t Nb, int D) {
    if (h->opt.score_f32mfma && !h->x3) return SCORE_F32MFMA;
    if (!h->opt.score_tiled && score_h3w_supported(D, Na, Nb) && ((reinterpret_cast<uintptr_t>(dA) | reinterpret_cast<uintptr_t>(dB)) & 15) == 0) return SCORE_H3W;
    return SCORE_WORDS;
}

static int score_gemm(svhip_handle* h, const char* label, const float* dA, int64_t Na, const float* dB, int64_t Nb, int D, float* dO, int64_t ldo,
                      ScoreRoute route, const void* dBsplit) {
    if (D % 32 != 0) SV_FAIL(h, SVHIP_ERR_UNSUPPORTED, "embedding dim %d must be a multiple of 32", D);
    if (Na > (1 << 30) / 1 || Nb > (1 << 30)) SV_FAIL(h, SVHIP_ERR_INVALID, "matrix too large");
    hipStream_t st = h->stream;
    if (route == SCORE_H3W) {
        if (!dBsplit || (reinterpret_cast<uintptr_t>(dA) & 15)) SV_FAIL(h, SVHIP_ERR_STATE, "score route: the row-streaming kernel was chosen without its plane scratch");
        return run(h, label, 2.0 * Na * Nb * D, [&]() { return launch_score_h3w(dA, Na, dB, Nb, D, dO, ldo, const_cast<void*>(dBsplit), h->num_cu, st); });
    }
    GemmParams p;
    p.A = dA; p.W = dB; p.Y = dO;
    p.M = (int)Na; p.N = (int)Nb; p.K = D; p.Kp = D; p.Wrows = (int)Nb;
    p.lda = D; p.ldy = (int)ldo; p.T = 1;
    if (route == SCORE_WORDS && dBsplit && gemm_pw_supported(p, false)) { p.W = dBsplit; p.x3 = 1; }      // (the words exist: split_b ran launch_split_words)
    return run(h, label, 2.0 * Na * Nb * D, [&]() { return launch_gemm(p, false, st); });
}

// the split operand of the chosen route in a scratch slot of the handle: half planes (filled by the row-streaming kernel's own launcher:
// [2][Nb][D] halves + its scale word) or split words (filled here)
static int split_b(svhip_handle* h, ScoreRoute route, const float* dB, int64_t Nb, int D, void** out) {
    *out = nullptr;
    if (route == SCORE_F32MFMA) return SVHIP_OK;
    if (int rc = scratch(h, svhip_handle::SCR_SPLIT, std::max((size_t)Nb * D * 4, score_h3w_planes_bytes(D, Nb)), out)) return rc;
    if (route == SCORE_H3W) return SVHIP_OK;
    return run(h, "split_words", 0, [&]() { return launch_split_words(dB, *out, Nb * D, h->stream); });
}

int svhip_score_matrix(svhip_handle* h, const float* A, int64_t Na, const float* B, int64_t Nb, int32_t D, float* out, int32_t flags) {
    if (!h || !A || !B || !out || Na <= 0 || Nb <= 0 || D <= 0) return SVHIP_ERR_INVALID;
    SV_HIP(h, hipSetDevice(h->cfg.device));
    const bool din = flags & SVHIP_IN_DEVICE, dout = flags & SVHIP_OUT_DEVICE;
    TempBuf tA{h, svhip_handle::SCR_IN0}, tB{h, svhip_handle::SCR_IN1}, tO{h, svhip_handle::SCR_OUT0};
    const void *dA, *dB;
    void* dO;
    int rc;
    if ((rc = tA.in(A, (size_t)Na * D * 4, din, &dA))) return rc;
    if ((rc = tB.in(B, (size_t)Nb * D * 4, din, &dB))) return rc;
    if ((rc = tO.out(out, (size_t)Na * Nb * 4, dout, &dO))) return rc;
    void* bsplit = nullptr;
    const ScoreRoute route = score_route(h, (const float*)dA, Na, (const float*)dB, Nb, D);
    if ((rc = split_b(h, route, (const float*)dB, Nb, D, &bsplit))) return rc;
    rc = score_gemm(h, "score_matrix", (const float*)dA, Na, (const float*)dB, Nb, D, (float*)dO, Nb, route, bsplit);
    if (!rc && !dout) { const hipError_t e = hipMemcpyAsync(out, dO, (size_t)Na * Nb * 4, hipMemcpyDeviceToHost, h->stream); if (e != hipSuccess) rc = SVHIP_ERR_HIP; }
    if (!(din && dout && (flags & SVHIP_ASYNC))) (void)hipStreamSynchronize(h->stream);
    return rc;
}

// the slab path: cohort scores of `rows` embeddings into an HBM scratch (rows x K fp32), slab by slab, reduced per row.
// Used for shapes the fused kernel does not take (small cohorts, top > 256, other embedding widths, F32X3 handles) and for
// the embeddings the fused kernel flags.
static int asnorm_stats_slab(svhip_handle* h, const float* dE, int64_t N, int D, const float* dC, int K, int top, float* dM, float* dS) {
    const int64_t ldk = (K + 3) & ~3;                             // row stride of the slab (16-byte rows for the DMA GEMM)
    const int64_t slab_rows = std::min<int64_t>(N, std::max<int64_t>(128, ((int64_t)1 << 31) / (ldk * 4)));
    void* csplit = nullptr;
    void* slab = nullptr;
    int rc;
    const ScoreRoute route = score_route(h, dE, slab_rows, dC, K, D);       // (every slab starts r0 * D * 4 bytes into E: the same alignment)
    if ((rc = split_b(h, route, dC, K, D, &csplit))) return rc;
    if ((rc = scratch(h, svhip_handle::SCR_SLAB, (size_t)slab_rows * ldk * 4, &slab))) return rc;
    for (int64_t r0 = 0; r0 < N; r0 += slab_rows) {
        const int64_t rows = std::min(slab_rows, N - r0);
        rc = score_gemm(h, "asnorm_cohort_gemm", dE + r0 * D, rows, dC, K, D, (float*)slab, ldk, route, csplit);
        if (!rc) rc = run(h, "asnorm_topk", 0, [&]() { return launch_topk_stats((const float*)slab, rows, K, (int)ldk, top, dM + r0, dS + r0, h->stream); });
        if (rc) return rc;
    }
    return SVHIP_OK;
}

int svhip_asnorm_stats(svhip_handle* h, const float* E, int64_t N, int32_t D, const float* cohort, int32_t K, int32_t top,
                       float* mu, float* sigma, int32_t flags) {
    if (!h || !E || !cohort || !mu || !sigma || N <= 0 || D <= 0 || K <= 0) return SVHIP_ERR_INVALID;
    if (top < 0) top = K + top;             // python slice semantics of S[:top] (utils.py:143); top=-1 drops the smallest
    if (top > K) top = K;
    if (top <= 0) SV_FAIL(h, SVHIP_ERR_INVALID, "top must select at least one cohort score");
    if (N >= ((int64_t)1 << 31)) SV_FAIL(h, SVHIP_ERR_UNSUPPORTED, "at most 2^31 - 1 embeddings per call");
    SV_HIP(h, hipSetDevice(h->cfg.device));
    const bool din = flags & SVHIP_IN_DEVICE, dout = flags & SVHIP_OUT_DEVICE;
    TempBuf tE{h, svhip_handle::SCR_IN0}, tC{h, svhip_handle::SCR_IN1}, tM{h, svhip_handle::SCR_OUT0}, tS{h, svhip_handle::SCR_OUT1};
    const void *dE, *dC;
    void *dM, *dS;
    int rc;
    if ((rc = tE.in(E, (size_t)N * D * 4, din, &dE))) return rc;
    if ((rc = tC.in(cohort, (size_t)K * D * 4, din, &dC))) return rc;
    if ((rc = tM.out(mu, (size_t)N * 4, dout, &dM))) return rc;
    if ((rc = tS.out(sigma, (size_t)N * 4, dout, &dS))) return rc;
    const bool aligned = ((reinterpret_cast<uintptr_t>(dE) | reinterpret_cast<uintptr_t>(dC)) & 15) == 0;
    if (!h->x3 && aligned && asnorm_fused_supported(D, K, top) && !h->opt.asnorm_slab) {
        // fused path: the scores never leave the MFMA accumulators (csrc/asnorm_fused.hip)
        // chunks of 131 072 embeddings: the candidate statistics of chunk c run on a second stream under the MFMA kernel of
        // chunk c + 1 (two candidate buffers; the kernels meet through events)
        const int64_t chunk = std::min<int64_t>(N, 131072);
        void *mb, *cand, *cnt, *flag;
        const size_t mb_bytes = (size_t)(D + 32) * D * 4;          // [MB | slice partials of its computation]
        const size_t cand_elems = (size_t)chunk * 2 * ASNORM_CAND_PER_LANE, cnt_elems = (size_t)chunk * 4;      // (2 or 4 candidate lists per embedding)
        // (the exact default is the six-bf16-MFMA form where it is built: scores to fp32 rounding at 2.7 x the fp32 matrix rate)
        // (the exact default is a split form where it is built: scores to fp32 rounding at several times the fp32 matrix rate — two half
        //  planes / three fp16 MFMAs (D = 192, 256); option asnorm_x6: three bf16 planes / six bf16 MFMAs, round 3's form, D = 192)
        const int nplanes = h->opt.asnorm_x6 ? 3 : 2;
        const bool x6 = asnorm_fused6_supported(D, nplanes) && !h->opt.asnorm_f32mfma;
        // (x6: the candidate kernel takes 1.7 ms of 17 on its own and 7 when it shares the CUs with the matrix kernel: one stream.
        //  The fp32-MFMA form keeps the second stream: 26.1 - 26.9 against 27.5 ms)
        const int nbuf = (N > chunk && (!x6 || h->opt.asnorm_2s)) ? 2 : 1;
        const size_t mom_bytes = (cohort_moments_scratch_bytes(D) + 255) & ~(size_t)255;
        if ((rc = scratch(h, svhip_handle::SCR_MB, mb_bytes + mom_bytes + (x6 ? asnorm_planes_bytes(D, K) : 0), &mb))) return rc;
        if ((rc = scratch(h, svhip_handle::SCR_CAND, cand_elems * 4 * nbuf, &cand))) return rc;
        if ((rc = scratch(h, svhip_handle::SCR_CNT, (cnt_elems + (size_t)chunk) * 4 * nbuf, &cnt))) return rc;      // per buffer: [counts (chunk, 4) | row factors (chunk)]
        if ((rc = scratch(h, svhip_handle::SCR_FLAG, (size_t)(2 * N + 1) * 4, &flag))) return rc;        // [count | flagged ids (N) | their candidate counts (N)]
        if (nbuf == 2 && !h->aux_stream) {
            SV_HIP(h, hipStreamCreateWithFlags(&h->aux_stream, hipStreamNonBlocking));
            for (int i = 0; i < 4; ++i) SV_HIP(h, hipEventCreateWithFlags(&h->aux_ev[i], hipEventDisableTiming));
        }
        int32_t* nflag = (int32_t*)flag;                    // [0] = number of flagged embeddings, [1 ..] their indices
        SV_HIP(h, hipMemsetAsync(nflag, 0, 4, h->stream));
        if ((rc = run(h, "asnorm_cohort_moments", 0, [&]() { return launch_cohort_moments((const float*)dC, K, D, (float*)mb, (float*)((char*)mb + mb_bytes), h->stream); }))) return rc;
        AsnormFusedParams fp;
        fp.cohort = (const float*)dC; fp.K = K; fp.MB = (const float*)mb; fp.z = asnorm_tail_z(K, top);
        if (x6) {
            void* planes = (char*)mb + mb_bytes + mom_bytes;
            fp.planes = planes; fp.nplanes = nplanes;
            fp.nlists = (nplanes == 2 && !h->opt.asnorm_w32) ? 4 : 2;       // the 16-wide-MFMA kernel: four lists per embedding
            // the default kernel scales its operands by exact powers of two (asnorm_fused.hip, "operand scaling"): the cohort's max |x| goes to
            // a device word behind the two planes (the buffer is sized for three)
            uint32_t* pscale = fp.nlists == 4 ? reinterpret_cast<uint32_t*>((char*)planes + (size_t)2 * (D + 32 + K) * D * 2) : nullptr;
            if ((rc = run(h, "asnorm_planes", 0, [&]() { return launch_asnorm_planes((const float*)mb, (const float*)dC, K, D, planes, h->stream, nplanes, pscale); }))) return rc;
            fp.pscale = pscale;
        }
        int c = 0;
        for (int64_t r0 = 0; r0 < N; r0 += chunk, ++c) {
            const int64_t rows = std::min(chunk, N - r0);
            const int b = c & (nbuf - 1);
            fp.E = (const float*)dE + r0 * D; fp.N = rows;
            fp.cand = (float*)cand + b * cand_elems; fp.cnt = (int32_t*)cnt + b * (cnt_elems + (size_t)chunk);
            fp.rowscale = fp.pscale ? reinterpret_cast<float*>(fp.cnt + cnt_elems) : nullptr;
            if (nbuf == 2 && c >= 2) SV_HIP(h, hipStreamWaitEvent(h->stream, h->aux_ev[2 + b], 0));       // the statistics of chunk c - 2 have read this buffer
            if ((rc = run(h, "asnorm_fused", 2.0 * rows * K * D, [&]() { return launch_asnorm_fused(fp, D, h->stream); }))) {
                if (nbuf == 2 && h->aux_stream) (void)hipStreamSynchronize(h->aux_stream);
                return rc;
            }
            hipStream_t st2 = h->stream;
            if (nbuf == 2) {
                SV_HIP(h, hipEventRecord(h->aux_ev[b], h->stream));
                SV_HIP(h, hipStreamWaitEvent(h->aux_stream, h->aux_ev[b], 0));
                st2 = h->aux_stream;
            }
            h->cur = st2;
            rc = run(h, "asnorm_cand_stats", 0, [&]() {
                return launch_asnorm_cand_stats(fp.cand, fp.cnt, rows, top, (float*)dM, (float*)dS, r0, nflag + 1, nflag, st2, fp.nlists, fp.rowscale, nflag + 1 + N);
            });
            h->cur = h->stream;
            if (rc) {           // leave no aux-stream work pending behind a failed call
                if (nbuf == 2) (void)hipStreamSynchronize(h->aux_stream);
                return rc;
            }
            if (nbuf == 2) SV_HIP(h, hipEventRecord(h->aux_ev[2 + b], h->aux_stream));
        }
        if (nbuf == 2) for (int b = 0; b < std::min(c, 2); ++b) SV_HIP(h, hipStreamWaitEvent(h->stream, h->aux_ev[2 + b], 0));
        int32_t nf = 0;
        SV_HIP(h, hipMemcpyAsync(&nf, nflag, 4, hipMemcpyDeviceToHost, h->stream));
        SV_HIP(h, hipStreamSynchronize(h->stream));
        int64_t refit_rows = 0;
        int refit_passes = 0;
        if (nf > 0 && x6 && fp.nlists == 4 && !h->opt.asnorm_norefit) {
            // REFIT (round 6).  tau = mean + z sd with z the normal quantile fits isotropic embeddings; real cohorts are not isotropic
            // (speaker centroids cluster by gender / language: bimodal cohort scores), and a row whose threshold passes fewer than `top`
            // scores, or overflows a list, used to take the slab path — N x K scores through HBM.  Such rows now go through the SAME fused
            // kernel again with a z of their own, derived from what the last pass counted (refit_next_z, asnorm_fused.hip).  At most ASNORM_REFIT_PASSES
            // passes; what is still undecided after them takes the slab path as before.
            constexpr int ASNORM_REFIT_PASSES = 3;
            const float target = fminf(1.6f * (float)top, 0.7f * (float)(2 * ASNORM_CAND_PER_LANE));
            // everything lives on the device: the state of the undecided rows (asnorm_fused.hip, "refit state": ids, z, the last measurement, the
            // bracket), the gathered rows, their statistics and the flag list of the pass; the host reads ONE word per pass (how many are left)
            void* g;
            const size_t cap = (size_t)nf;
            const size_t gbytes = cap * D * 4 + cap * 4 * (2 + 6 + 6 + 2) + 256;
            if ((rc = scratch(h, svhip_handle::SCR_GATHER, gbytes, &g))) return rc;
            float* gE = (float*)g;
            float* gM = gE + cap * D;
            float* gS = gM + cap;
            float* soa[2] = {gS + cap, gS + cap + 6 * cap};
            int32_t* gF = (int32_t*)(soa[1] + 6 * cap);      // [0] = count, [1 ..] positions within the gathered list
            int32_t* gInfo = gF + 1 + cap;                   // (gbytes: + 256 covers the counter word)
            if ((rc = run(h, "asnorm_refit_state", 0, [&]() { return launch_asnorm_refit_init(nflag + 1, nflag + 1 + N, nf, fp.z, target, soa[0], h->stream); }))) return rc;
            int cur = nf, a = 0;
            while (cur > 0 && refit_passes < ASNORM_REFIT_PASSES) {
                ++refit_passes;
                const int32_t* gI = reinterpret_cast<const int32_t*>(soa[a]);
                const float* gZ = soa[a] + cur;
                SV_HIP(h, hipMemsetAsync(gF, 0, 4, h->stream));
                if ((rc = run(h, "asnorm_gather", 0, [&]() { return launch_gather_rows((const float*)dE, gI, cur, D, gE, h->stream); }))) return rc;
                for (int64_t r0 = 0; r0 < cur; r0 += chunk) {
                    const int64_t rows = std::min<int64_t>(chunk, cur - r0);
                    fp.E = gE + r0 * D; fp.N = rows; fp.zrow = gZ + r0;
                    fp.cand = (float*)cand; fp.cnt = (int32_t*)cnt;
                    fp.rowscale = fp.pscale ? reinterpret_cast<float*>(fp.cnt + cnt_elems) : nullptr;
                    if ((rc = run(h, "asnorm_fused_refit", 2.0 * rows * K * D, [&]() { return launch_asnorm_fused(fp, D, h->stream); }))) return rc;
                    if ((rc = run(h, "asnorm_cand_stats", 0, [&]() {
                            return launch_asnorm_cand_stats(fp.cand, fp.cnt, rows, top, gM, gS, r0, gF + 1, gF, h->stream, fp.nlists, fp.rowscale, gInfo);
                        }))) return rc;
                }
                fp.zrow = nullptr;
                // (rows that are still undecided scatter whatever their slots hold: a later pass or the slab path overwrites them)
                if ((rc = run(h, "asnorm_scatter", 0, [&]() { return launch_scatter_stats(gM, gS, gI, cur, (float*)dM, (float*)dS, h->stream); }))) return rc;
                int32_t left = 0;
                SV_HIP(h, hipMemcpyAsync(&left, gF, 4, hipMemcpyDeviceToHost, h->stream));
                SV_HIP(h, hipStreamSynchronize(h->stream));
                refit_rows += cur - left;
                // (the flag order is whatever the atomics gave; a row's result does not depend on where it sits in the gathered list)
                if (left > 0 && (rc = run(h, "asnorm_refit_state", 0, [&]() {
                                     return launch_asnorm_refit_next(soa[a], cur, gF + 1, gInfo, left, target, soa[a ^ 1], h->stream);
                                 }))) return rc;
                a ^= 1;
                cur = left;
            }
            // what the refit passes could not decide: back into the flag list for the slab path
            nf = cur;
            if (nf > 0) SV_HIP(h, hipMemcpyAsync(nflag + 1, soa[a], (size_t)nf * 4, hipMemcpyDeviceToDevice, h->stream));
        }
        h->last_asnorm_refit = refit_rows;
        h->last_asnorm_refit_passes = refit_passes;
        if (nf > 0) {
            // cohort scores too far from normal for the threshold (fewer than `top` candidates, or a list overflowed) even after the refit
            // passes: these embeddings are gathered and take the slab path; results are scattered back
            void* g;
            const size_t gbytes = (size_t)nf * D * 4 + 2 * (size_t)nf * 4 + 64;
            if ((rc = scratch(h, svhip_handle::SCR_GATHER, gbytes, &g))) return rc;
            float* gE = (float*)g;
            float* gM = gE + (size_t)nf * D;
            float* gS = gM + nf;
            if ((rc = run(h, "asnorm_gather", 0, [&]() { return launch_gather_rows((const float*)dE, nflag + 1, nf, D, gE, h->stream); }))) return rc;
            if ((rc = asnorm_stats_slab(h, gE, nf, D, (const float*)dC, K, top, gM, gS))) return rc;
            if ((rc = run(h, "asnorm_scatter", 0, [&]() { return launch_scatter_stats(gM, gS, nflag + 1, nf, (float*)dM, (float*)dS, h->stream); }))) return rc;
        }
        h->last_asnorm_flagged = nf;
    } else {
        if ((rc = asnorm_stats_slab(h, (const float*)dE, N, D, (const float*)dC, K, top, (float*)dM, (float*)dS))) return rc;
        h->last_asnorm_flagged = -1;
        h->last_asnorm_refit = 0; h->last_asnorm_refit_passes = 0;
    }
    if (!dout) {
        SV_HIP(h, hipMemcpyAsync(mu, dM, (size_t)N * 4, hipMemcpyDeviceToHost, h->stream));
        SV_HIP(h, hipMemcpyAsync(sigma, dS, (size_t)N * 4, hipMemcpyDeviceToHost, h->stream));
    }
    if (!(din && dout && (flags & SVHIP_ASYNC))) {
        SV_HIP(h, hipStreamSynchronize(h->stream));
        release_big_scratch(h, !din || !dout);
    }
    return SVHIP_OK;
}

int64_t svhip_asnorm_last_fallback(const svhip_handle* h) { return h ? h->last_asnorm_flagged : -1; }
int64_t svhip_asnorm_last_refit(const svhip_handle* h, int32_t* passes) {
    if (passes) *passes = h ? h->last_asnorm_refit_passes : 0;
    return h ? h->last_asnorm_refit : 0;
}

// ---- introspection ---------------------------------------------------------------------------------------
// ---- verification metrics (metrics.hip) ------------------------------------------------------------------------------------
namespace {
// shared front half: stage scores / labels, sort, scan.  Labels must be 0 / 1 (host labels are checked).
struct MetricsRun {
    svhip_handle* h;
    TempBuf tS{nullptr, svhip_handle::SCR_IN0}, tL{nullptr, svhip_handle::SCR_IN1};
    void* ws = nullptr;
    size_t ws_bytes = 0;
    explicit MetricsRun(svhip_handle* hh) : h(hh) { tS.h = hh; tL.h = hh; }
    int start(const float* scores, const int32_t* labels, int64_t P, bool din, bool nan_to_num, const char* label) {
        if (P >= ((int64_t)1 << 31)) SV_FAIL(h, SVHIP_ERR_UNSUPPORTED, "trial lists are limited to 2^31 - 1 entries");
        if (!din)
            for (int64_t i = 0; i < P; ++i)
                if (labels[i] != 0 && labels[i] != 1) SV_FAIL(h, SVHIP_ERR_INVALID, "label %lld of trial %lld is not 0 / 1", (long long)labels[i], (long long)i);
        SV_HIP(h, hipSetDevice(h->cfg.device));
        const void *dS, *dL;
        int rc;
        if ((rc = tS.in(scores, (size_t)P * 4, din, &dS))) return rc;
        if ((rc = tL.in(labels, (size_t)P * 4, din, &dL))) return rc;
        ws_bytes = metrics_workspace_bytes(P);
        if ((rc = scratch(h, svhip_handle::SCR_WS, ws_bytes, &ws))) return rc;
        return run(h, label, 0, [&]() { return metrics_sort_scan((const float*)dS, (const int32_t*)dL, P, nan_to_num, ws, ws_bytes, h->stream); });
    }
};
}  // namespace

int svhip_roc_points(svhip_handle* h, const float* scores, const int32_t* labels, int64_t P, int64_t* n_out, float* thr, int64_t* fps,
                     int64_t* tps, int32_t flags) {
    if (!h || !scores || !labels || !n_out || !thr || !fps || !tps || P <= 0) return SVHIP_ERR_INVALID;
    const bool din = flags & SVHIP_IN_DEVICE, dout = flags & SVHIP_OUT_DEVICE;
    MetricsRun m(h);
    int rc;
    if ((rc = m.start(scores, labels, P, din, true, "metrics_sort"))) return rc;
    TempBuf tT{h, svhip_handle::SCR_OUT0}, tF{h, svhip_handle::SCR_OUT1}, tP{h, svhip_handle::SCR_OUT2};
    void *dT, *dF, *dP;
    if ((rc = tT.out(thr, (size_t)P * 4, dout, &dT)) || (rc = tF.out(fps, (size_t)P * 8, dout, &dF)) || (rc = tP.out(tps, (size_t)P * 8, dout, &dP))) return rc;
    int32_t* n_dev = nullptr;
    if ((rc = run(h, "metrics_roc", 0, [&]() { return metrics_roc_points(P, m.ws, m.ws_bytes, (float*)dT, (int64_t*)dF, (int64_t*)dP, &n_dev, h->stream); }))) return rc;
    int32_t n32 = 0;
    SV_HIP(h, hipMemcpyAsync(&n32, n_dev, 4, hipMemcpyDeviceToHost, h->stream));
    SV_HIP(h, hipStreamSynchronize(h->stream));
    *n_out = n32;
    if (!dout) {
        SV_HIP(h, hipMemcpy(thr, dT, (size_t)n32 * 4, hipMemcpyDeviceToHost));
        SV_HIP(h, hipMemcpy(fps, dF, (size_t)n32 * 8, hipMemcpyDeviceToHost));
        SV_HIP(h, hipMemcpy(tps, dP, (size_t)n32 * 8, hipMemcpyDeviceToHost));
    }
    return SVHIP_OK;
}

int svhip_error_rates(svhip_handle* h, const float* scores, const int32_t* labels, int64_t P, double* fnrs, double* fprs, float* thresholds,
                      int32_t flags) {
    if (!h || !scores || !labels || !fnrs || !fprs || !thresholds || P <= 0) return SVHIP_ERR_INVALID;
    const bool din = flags & SVHIP_IN_DEVICE, dout = flags & SVHIP_OUT_DEVICE;
    MetricsRun m(h);
    int rc;
    if ((rc = m.start(scores, labels, P, din, false, "metrics_sort"))) return rc;
    TempBuf tA{h, svhip_handle::SCR_OUT0}, tB{h, svhip_handle::SCR_OUT1}, tC{h, svhip_handle::SCR_OUT2};
    void *dA, *dB, *dC;
    if ((rc = tA.out(fnrs, (size_t)P * 8, dout, &dA)) || (rc = tB.out(fprs, (size_t)P * 8, dout, &dB)) || (rc = tC.out(thresholds, (size_t)P * 4, dout, &dC))) return rc;
    if ((rc = run(h, "metrics_rates", 0, [&]() { return metrics_error_rates(P, m.ws, m.ws_bytes, (double*)dA, (double*)dB, (float*)dC, h->stream); }))) return rc;
    if (!dout) {
        SV_HIP(h, hipMemcpyAsync(fnrs, dA, (size_t)P * 8, hipMemcpyDeviceToHost, h->stream));
        SV_HIP(h, hipMemcpyAsync(fprs, dB, (size_t)P * 8, hipMemcpyDeviceToHost, h->stream));
        SV_HIP(h, hipMemcpyAsync(thresholds, dC, (size_t)P * 4, hipMemcpyDeviceToHost, h->stream));
    }
    SV_HIP(h, hipStreamSynchronize(h->stream));
    return SVHIP_OK;
}

int svhip_min_dcf(svhip_handle* h, const float* scores, const int32_t* labels, int64_t P, double p_target, double c_miss, double c_fa,
                  double* min_dcf, float* threshold, int32_t flags) {
    if (!h || !scores || !labels || !min_dcf || !threshold || P <= 0) return SVHIP_ERR_INVALID;
    MetricsRun m(h);
    int rc;
    if ((rc = m.start(scores, labels, P, flags & SVHIP_IN_DEVICE, false, "metrics_sort"))) return rc;
    void* res = nullptr;
    if ((rc = scratch(h, svhip_handle::SCR_OUT0, 16, &res))) return rc;
    rc = run(h, "metrics_min_dcf", 0, [&]() { return metrics_min_dcf(P, m.ws, m.ws_bytes, p_target, c_miss, c_fa, (double*)res, (float*)((char*)res + 8), h->stream); });
    char host[16];
    hipError_t e = rc ? hipSuccess : hipMemcpyAsync(host, res, 16, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (rc) return rc;
    SV_HIP(h, e);
    memcpy(min_dcf, host, 8);
    memcpy(threshold, host + 8, 4);
    return SVHIP_OK;
}

int svhip_get_stage(svhip_handle* h, const char* name, float* out, int64_t* count) {
    if (!h || !name || !count) return SVHIP_ERR_INVALID;
    if (h->lastB <= 0) SV_FAIL(h, SVHIP_ERR_STATE, "no forward has run yet");
    const int B = h->lastB, T = h->T, C = h->cfg.channels, C3 = 3 * C, e = h->esz;
    const size_t M = (size_t)B * T;
    const void* src = nullptr;
    size_t rows = M, cols = 0, ld = 0;
    bool f32 = !h->bf16;
    const std::string n(name);
    if (n == "input") { src = h->X_in; cols = ld = h->cfg.n_mels; }
    else if (n == "blocks.0") {
        src = h->X0; cols = ld = C;
        if (h->x0_is_s32 && out) {            // F32X3: X0 holds hi | lo planes; the fp32 view goes to the (idle) operand staging buffer
            SV_HIP(h, launch_unsplit_s32(h->X0, C, static_cast<float*>(h->s32_buf), C, (int64_t)M, C, h->stream));
            src = h->s32_buf;
        }
    }
    else if (n == "blocks.1" || n == "blocks.2" || n == "blocks.3") {
        const int i = n.back() - '1';
        src = off(h->CAT, (size_t)i * C, e); cols = C; ld = C3;
        if (h->cat_f32_stale && out) {        // F32X3: the block outputs exist only in the split layout; rebuild the fp32 view
            SV_HIP(h, launch_unsplit_s32(h->cat_s32, C3, static_cast<float*>(h->CAT), C3, (int64_t)M, C3, h->stream));
            h->cat_f32_stale = false;
        }
    }
    else if (n == "mfa") { src = h->MFA; cols = ld = C3; }
    else if (n == "asp") { src = h->d_pool_raw; rows = B; cols = ld = 2 * C3; f32 = true; }
    else if (n == "asp_bn") { src = h->d_pool_bn; rows = B; cols = ld = 2 * C3; f32 = true; }
    else if (n == "rn_x") { src = h->rn_dbg_x; rows = (size_t)B * h->rn_dbg_T; cols = ld = h->rn_dbg_C; }
    else if (n == "rn_snap") { src = h->rn_snap; rows = (size_t)B * h->rn_snap_T; cols = ld = h->rn_snap_C; }
    else if (n == "rn_pooled") { src = h->rn_pooled; rows = B; cols = ld = 1024; f32 = true; }
    else if (n == "mel") {
        if (h->feat_is_stale) SV_FAIL(h, SVHIP_ERR_STATE, "stage mel: the last forward ran the fused front-end, which never forms the mel power "
                                      "tensor (option fbank_unfused = 1 keeps the separate kernels)");
        src = h->d_feat; rows = (size_t)B * h->cfg.n_mels; cols = ld = T; f32 = true;
    }
    else SV_FAIL(h, SVHIP_ERR_INVALID, "unknown stage %s", name);
    *count = (int64_t)(rows * cols);
    if (!out) return SVHIP_OK;
    SV_HIP(h, hipStreamSynchronize(h->stream));
    const size_t es = f32 ? 4 : 2;
    std::vector<char> tmp(rows * cols * es);
    SV_HIP(h, hipMemcpy2D(tmp.data(), cols * es, src, ld * es, cols * es, rows, hipMemcpyDeviceToHost));
    if (f32) memcpy(out, tmp.data(), tmp.size());
    else {
        const uint16_t* s = reinterpret_cast<const uint16_t*>(tmp.data());
        if (h->f16) for (size_t i = 0; i < rows * cols; ++i) { _Float16 hv; memcpy(&hv, &s[i], 2); out[i] = static_cast<float>(hv); }
        else for (size_t i = 0; i < rows * cols; ++i) { uint32_t u = (uint32_t)s[i] << 16; memcpy(&out[i], &u, 4); }
    }
    return SVHIP_OK;
}

int svhip_profile_enable(svhip_handle* h, int32_t on) { if (!h) return SVHIP_ERR_INVALID; h->prof = on != 0; return SVHIP_OK; }
int svhip_profile_filter(svhip_handle* h, const char* label) { if (!h) return SVHIP_ERR_INVALID; h->prof_filter = label ? label : ""; return SVHIP_OK; }
int svhip_profile_reset(svhip_handle* h) { if (!h) return SVHIP_ERR_INVALID; prof_collect(h); h->prof_entries.clear(); return SVHIP_OK; }
int svhip_profile_get(svhip_handle* h, int32_t idx, char* name, int32_t name_cap, double* ms, int64_t* launches, double* flops) {
    if (!h) return SVHIP_ERR_INVALID;
    prof_collect(h);
    if (idx < 0 || idx >= (int32_t)h->prof_entries.size()) return SVHIP_ERR_INVALID;
    const ProfEntry& p = h->prof_entries[idx];
    if (name && name_cap > 0) { strncpy(name, p.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
    if (ms) *ms = p.ms;
    if (launches) *launches = p.launches;
    if (flops) *flops = p.flops;
    return SVHIP_OK;
}
double svhip_workload_flops(const svhip_handle* h) { return h ? h->flops_per_utt : 0.0; }

int svhip_set_option(svhip_handle* h, const char* name, int32_t value) {
    if (!h || !name) return SVHIP_ERR_INVALID;
    const std::string n(name);
    svhip_handle::DevOpts& o = h->opt;
    struct { const char* key; int* slot; } table[] = {
        {"layer_labels", &o.layer_labels}, {"x3_keep_f32", &o.x3_keep_f32}, {"r2_big", &o.r2_big}, {"asp_v1", &o.asp_v1},
        {"rn_stop", &o.rn_stop}, {"rn_snap", &o.rn_snap}, {"rn_unfused", &o.rn_unfused}, {"asnorm_slab", &o.asnorm_slab},
        {"asnorm_f32mfma", &o.asnorm_f32mfma}, {"asnorm_x6", &o.asnorm_x6}, {"asnorm_2s", &o.asnorm_2s}, {"asnorm_norefit", &o.asnorm_norefit}, {"asnorm_w32", &o.asnorm_w32}, {"score_tiled", &o.score_tiled}, {"score_f32mfma", &o.score_f32mfma}, {"fbank32", &o.fbank32}, {"rn_sinc_full", &o.rn_sinc_full}, {"fbank_unfused", &o.fbank_unfused}, {"ff_abl", &o.ff_abl}, {"pw3_cus", &o.pw3_cus}, {"pw3_tail_off", &o.pw3_tail_off}, {"pw4", &o.pw4}, {"cv_off", &o.cv_off},
        {"r2_slices", &o.r2_slices}, {"rn_tail_big", &o.rn_tail_big}, {"rn_sinc_f32", &o.rn_sinc_f32}, {"rn_step_off", &o.rn_step_off}, {"rn_pool_off", &o.rn_pool_off}, {"n128_off", &o.n128_off}};
    for (auto& t : table)
        if (n == t.key) {
            *t.slot = value;
            h->fb.force32 = o.fbank32; h->fb.ff_abl = o.ff_abl;
            return SVHIP_OK;
        }
    SV_FAIL(h, SVHIP_ERR_INVALID, "unknown option %s", name);
}

// Release the scoring / metrics scratch slots (they are grown on demand and otherwise kept for the life of the handle: one slab-path
// AS-norm call or one large host-pointer call would hold gigabytes of HBM beside the model engines' workspaces).
int svhip_trim_scratch(svhip_handle* h) {
    if (!h) return SVHIP_ERR_INVALID;
    SV_HIP(h, hipSetDevice(h->cfg.device));
    SV_HIP(h, hipStreamSynchronize(h->stream));
    if (h->aux_stream) SV_HIP(h, hipStreamSynchronize(h->aux_stream));
    for (int i = 0; i < svhip_handle::SCR_COUNT; ++i)
        if (h->scr[i]) { (void)hipFree(h->scr[i]); h->scr[i] = nullptr; h->scr_cap[i] = 0; }
    return SVHIP_OK;
}

// host-only self checks (no GPU needed): the per-device one-time flag every LDS-hungry launcher keeps
int svhip_selftest(void) {
    svhip::DeviceOnce once;
    for (int d = -1; d < 66; ++d) if (once.done(d)) return 1;                 // nothing marked yet, out-of-range ordinals never are
    once.mark(0);
    if (!once.done(0) || once.done(1) || once.done(63)) return 2;            // device 0 set up says nothing about device 1
    once.mark(5); once.mark(63); once.mark(64); once.mark(-3);                // out-of-range marks are ignored
    if (!once.done(5) || !once.done(63) || once.done(64) || once.done(-3) || once.done(4)) return 3;
    once.mark(0);
    if (!once.done(0) || once.mask.load() != ((1ull << 0) | (1ull << 5) | (1ull << 63))) return 4;
    return 0;
}

}  // extern "C"
