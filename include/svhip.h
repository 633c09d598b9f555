/*
 * svhip.h — C ABI of the MI355X-native speaker-embedding + scoring hot path (libsvhip.so).
 *
 * This is the drop-in boundary: plain pointers and sizes, no torch / C++ types.
 * Every entry point names the reference interface it replaces (paths relative to the reference
 * checkout, hiimmuc/SpeakerVerification).  The reference is pure Python, so the "FFI" a maintainer
 * adds is a ctypes binding — see INTEGRATION.md and speakerverification_amd/_lib.py.
 *
 * Conventions
 *   - every function returns an svhip_status (0 = ok, negative = error) and never throws;
 *     svhip_last_error(h) returns the message of the last failing call on that handle.
 *   - one handle = one HIP device + one HIP stream + one model's device weights + workspace.
 *     Handles are thread-compatible, not thread-safe (one host thread per handle).
 *   - the caller owns every buffer it passes.  `flags` says where they live:
 *     SVHIP_IN_DEVICE / SVHIP_OUT_DEVICE mark device pointers; otherwise host pointers are
 *     staged through the handle's own device buffers.  With SVHIP_ASYNC the call returns after
 *     enqueueing on the handle's stream (device pointers only); otherwise it synchronises.
 *   - activations inside the library are frame-major (B, T, C); the boundary keeps the
 *     reference's layouts: waveforms (B, L) fp32, features (B, n_mels, T) fp32, embeddings
 *     (B, nOut) fp32, all contiguous.
 */
#ifndef SVHIP_H
#define SVHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVHIP_ABI_VERSION 5

typedef struct svhip_handle svhip_handle;

typedef enum svhip_status {
    SVHIP_OK = 0,
    SVHIP_ERR_INVALID = -1,     /* bad argument / shape */
    SVHIP_ERR_HIP = -2,         /* a HIP runtime call failed */
    SVHIP_ERR_STATE = -3,       /* call order violated (e.g. embed before finalize) */
    SVHIP_ERR_NOMEM = -4,
    SVHIP_ERR_UNSUPPORTED = -5,
    SVHIP_ERR_MISSING = -6,     /* a required weight tensor was never loaded */
    SVHIP_ERR_NONFINITE = -7,   /* the call completed and wrote its embeddings, but some of them are inf / NaN: an fp16 activation
                                   overflowed (SVHIP_F16 stores activations as IEEE half: 65504), or the input was not finite */
    SVHIP_ERR_RANGE = -8        /* the call completed, but input values left the range the handle's arithmetic represents (SVHIP_F32X3:
                                   operands travel as IEEE-half hi | lo planes, |x| <= 65504): such values become inf in the planes and the
                                   embeddings come out non-finite */
} svhip_status;

enum { SVHIP_MODEL_ECAPA = 0, SVHIP_MODEL_RAWNET2 = 1, SVHIP_MODEL_NONE = 2 /* fbank + scoring only */ };
enum { SVHIP_F32 = 0, SVHIP_BF16 = 1, SVHIP_I64 = 2, SVHIP_F32X3 = 3 /* compute only */, SVHIP_F16 = 4 /* compute only */ };
enum { SVHIP_IN_DEVICE = 1, SVHIP_OUT_DEVICE = 2, SVHIP_ASYNC = 4 };

typedef struct svhip_config {
    int32_t struct_size;    /* = sizeof(svhip_config), for ABI evolution */
    int32_t model;          /* SVHIP_MODEL_* */
    int32_t compute;        /* SVHIP_F32: fp32 MFMA, 1e-4 parity path; SVHIP_BF16: bf16 MFMA, fp32 accumulate;
                               SVHIP_F32X3: fp32 storage and arithmetic everywhere except the convolution GEMMs, whose products are three
                               fp16 MFMAs on operands split into IEEE-half hi | lo parts (11 + 11 significant bits: 2^-23 relative while
                               |x| >= 2^-3, at most 2^-25 absolute below that): 1e-4 parity at ~3x the fp32 speed.  RANGE CONTRACT: an operand
                               of such a GEMM must satisfy |x| <= 65504 (the conversion to the planes is the plain one: beyond that hi is inf
                               and the embeddings come out inf / NaN).  Activations behind a BatchNorm are O(10); the network INPUT is checked on every
                               call (SVHIP_ERR_RANGE), the embeddings are checked for inf / NaN on every call (SVHIP_ERR_NONFINITE);
                               SVHIP_F16 (RawNet2 handles only): fp16 storage + fp16 MFMA, fp32 accumulate — the same speed as bf16 with
                               three more mantissa bits (RawNet2's un-normalised residual stack loses two digits to bf16 WEIGHT rounding);
                               an fp16 value overflows at 65504: the input is LayerNorm'ed, so only the weights decide the activation scale */
    int32_t device;         /* HIP device ordinal */
    int32_t channels;       /* ECAPA C (channels = [C,C,C,C,3C], ECAPA_TDNN.py:378) */
    int32_t n_mels;         /* 80 */
    int32_t embed_dim;      /* nOut: 192 (ECAPA) / 320 (RawNet2) */
    int32_t max_batch;      /* workspace is sized for max_batch utterances per call */
    int32_t samples;        /* L: samples per utterance (32000); fixes T = L/hop + 1 */
    int32_t log_input;      /* ECAPA: 1 = features=='melspectrogram' -> log(x+1e-6) - mean_t (ECAPA_TDNN.py:473-476) */
    int32_t input_norm;     /* ECAPA: InstanceNorm1d(n_mels, affine) (ECAPA_TDNN.py:406-409,477-478) */
    /* mel front-end — defaults of models/FeatureExtraction/feature.py:66-71 */
    int32_t fb_sr;          /* 8000 */
    int32_t n_fft;          /* 512 */
    int32_t win_length;     /* 200 */
    int32_t hop_length;     /* 80 */
    float   fmin;           /* 0 */
    float   fmax;           /* <=0 -> sr/2 */
    float   preemph;        /* 0.97; <0 disables pre-emphasis */
    void*   stream;         /* optional hipStream_t to enqueue on (NULL: the handle creates its own) */
} svhip_config;

/* Fill *cfg with the reference defaults (ECAPA C=1024, fp32, 80 mels, nOut 192, L=32000). */
void svhip_default_config(svhip_config* cfg);
int  svhip_abi_version(void);

/* Lifetime.  Replaces: SpeakerEncoder.__init__ building compute_features + __S__ on a device
 * (src/model.py:61-73).  */
int svhip_create(const svhip_config* cfg, svhip_handle** out);
int svhip_destroy(svhip_handle* h);
const char* svhip_last_error(const svhip_handle* h);   /* h may be NULL: last create() error */
int svhip_synchronize(svhip_handle* h);      /* also reports (and clears) the numeric status of the asynchronous calls it waited for */
/* Numeric status of the forwards since the last reset.  Every forward checks (a) its embeddings for inf / NaN (in the kernel that writes
 * them out) and (b), on SVHIP_F32X3 handles, its input features against the split planes' range.  A synchronous call returns
 * SVHIP_ERR_NONFINITE / SVHIP_ERR_RANGE itself (after writing its outputs); after SVHIP_ASYNC calls ask here (waits for the stream) or
 * call svhip_synchronize.  reset != 0 clears the status.  Returns SVHIP_OK, SVHIP_ERR_NONFINITE or SVHIP_ERR_RANGE (message: last_error). */
int svhip_numeric_status(svhip_handle* h, int32_t reset);

/* Weights.  Replaces: ModelHandling.loadParameters' name-matched state_dict copy
 * (src/model.py:718-746).  `name` is the reference state_dict key of the __S__ module (e.g.
 * "blocks.1.tdnn1.conv.conv.weight"); data is host memory, dtype SVHIP_F32 or SVHIP_I64
 * (num_batches_tracked, ignored).  Unknown names return SVHIP_ERR_INVALID, shape mismatches too —
 * the Python shim decides whether to skip them as the reference does.  finalize folds BatchNorm
 * into scale/shift, packs conv weights [N][tap][cin] (bf16 copy for the bf16 path), bakes the
 * sinc filters (RawNet_baseline.py:339-357) and uploads everything. */
int svhip_load_tensor(svhip_handle* h, const char* name, const void* data,
                      const int64_t* shape, int32_t ndim, int32_t dtype);
int svhip_finalize_weights(svhip_handle* h);

/* Feature front-end.  Replaces: compute_features(inp) = Sequential(PreEmphasis, nnAudio
 * MelSpectrogram) (models/FeatureExtraction/feature.py:66-94, src/utils.py:53-71;
 * call site src/model.py:112-113).  wav (B, L) fp32 -> mel power (B, n_mels, T) fp32,
 * T = L / hop + 1. */
int svhip_fbank(svhip_handle* h, const float* wav, int32_t B, int32_t L, float* mel_out, int32_t flags);

/* Model forward.  Replaces: self.__S__.forward(inp) (src/model.py:119-121) ==
 * ECAPA_TDNN.forward (models/ECAPA_TDNN.py:460-502) on features (B, n_mels, T), or
 * RawNet2.forward (models/RawNet2_custom.py:161-227) on waveforms (B, L).  emb_out (B, nOut) fp32
 * (the Python shim applies the reference's squeeze()). */
int svhip_embed_features(svhip_handle* h, const float* feat, int32_t B, int32_t T, float* emb_out, int32_t flags);
/* Fused waveform -> embedding: SpeakerEncoder.forward with label=None (src/model.py:104-125):
 * compute_features then __S__.forward (ECAPA), or __S__.forward directly (RawNet2). */
int svhip_embed_wave(svhip_handle* h, const float* wav, int32_t B, int32_t L, float* emb_out, int32_t flags);

/* Eval-mode cropping on device.  Replaces, for decoded 16-bit PCM, the cropping half of loadWAV
 * (src/processing/audio_loader.py:110-150): wrap-pad files not longer than L to L+1 samples, take num_eval
 * crops of L samples at int(linspace(0, len - L, num_eval)), scale by 1/32768 (soundfile float32).  pcm holds
 * the files back to back; file f is pcm[offsets[f] .. offsets[f] + lengths[f]).  crops_out is
 * (n_files * num_eval, L) fp32.  Ships int16 over PCIe instead of num_eval overlapping fp32 crops.  Host PCM is staged in a
 * buffer of the handle; with SVHIP_ASYNC (host or device input, device output) the call returns once the copy and the kernel
 * are enqueued — offsets / lengths are copied before it returns, the PCM array must stay valid until the stream has passed
 * the copy (immediately for pageable memory; a later synchronisation for pinned memory, which is what makes it overlap). */
int svhip_crop_pcm16(svhip_handle* h, const int16_t* pcm, int64_t n_samples, const int64_t* offsets, const int32_t* lengths,
                     int32_t n_files, int32_t num_eval, int32_t L, float* crops_out, int32_t flags);

/* Scoring.  Replaces the per-trial loop of ModelHandling.evaluateFromList / testFromList
 * (src/model.py:415-448,526-553) and src/utils.py:126-169.
 *   l2norm       : F.normalize(p=2, dim=1) in place (src/model.py:421-423), eps 1e-12.
 *   score_pairs  : out[p] = | cos(E[ia[p]], E[ib[p]]) |, per-norm clamp 1e-5 (utils.py:163-164,
 *                  one crop per row).
 *   score_matrix : out (Na, Nb) = A @ B^T (np.inner, utils.py:150).  On every handle the product runs as three fp16 MFMAs on IEEE-half
 *                  hi | lo planes of the operands (fp32-grade: ~1e-7 of |a||b| from the float64 product); both operands are first scaled
 *                  by exact powers of two on the device (a row of A by its own max |x|, B by its global max |x|), so for D in {192, 256}
 *                  (the row-streaming kernel; 16-byte aligned operands) ANY finite magnitude is served, and an inf / NaN element stays in
 *                  its own row or column (non-finite elements are left out of the max |x| searches).  Other widths run a tiled split GEMM
 *                  without the scaling: |x| <= 65504 there (beyond it the scores come out inf / NaN).  Option score_f32mfma keeps the
 *                  exact fp32 MFMA (any magnitude, any width).
 *   asnorm_stats : per row of E: S = cohort @ e (utils.py:142), top-`top` largest, population
 *                  mean / std (utils.py:143-146) -> mu[N], sigma[N].  For D in {192, 256}, top <= 256 and K >= 4 top the cohort
 *                  scores never reach memory (fused selection in the half-plane MFMA kernel, csrc/asnorm_fused.hip: the same three-MFMA
 *                  products and the same power-of-two operand scaling as score_matrix; option asnorm_f32mfma: exact fp32 MFMA); otherwise, and for
 *                  the embeddings that kernel cannot decide even with a threshold re-derived from its own counts (up to three more passes over those
 *                  embeddings only: svhip_asnorm_last_refit()), they exist only as <= 2 GiB slabs.  svhip_asnorm_last_fallback():
 *                  how many embeddings of the last call took the slab path after the fused kernel (-1: the whole call did).
 *   asnorm_pairs : out[p] = 0.5*((s-mu[a])/sd[a] + (s-mu[b])/sd[b]), s = E[a].E[b] (utils.py:148-160).
 * Pointers follow `flags` (indices are int32, device or host like the other inputs). */
int svhip_l2norm(svhip_handle* h, float* E, int64_t N, int32_t D, int32_t flags);
int svhip_score_pairs(svhip_handle* h, const float* E, int64_t N, int32_t D,
                      const int32_t* ia, const int32_t* ib, int64_t P, float* out, int32_t flags);
int svhip_score_matrix(svhip_handle* h, const float* A, int64_t Na, const float* B, int64_t Nb,
                       int32_t D, float* out, int32_t flags);
int svhip_asnorm_stats(svhip_handle* h, const float* E, int64_t N, int32_t D, const float* cohort,
                       int32_t K, int32_t top, float* mu, float* sigma, int32_t flags);
int svhip_asnorm_pairs(svhip_handle* h, const float* E, int64_t N, int32_t D, const float* mu,
                       const float* sigma, const int32_t* ia, const int32_t* ib, int64_t P,
                       float* out, int32_t flags);
int64_t svhip_asnorm_last_fallback(const svhip_handle* h);
/* ABI v5: embeddings of the last svhip_asnorm_stats call whose threshold the fused kernel re-derived from its own counts (cohort scores that
 * are not normally distributed: clustered speaker centroids) and decided in `*passes` extra passes (NULL: not wanted) instead of the slab path. */
int64_t svhip_asnorm_last_refit(const svhip_handle* h, int32_t* passes);

/* Whole-trial scores over the crops of two files (what ModelHandling.evaluateFromList needs per trial, src/model.py:413-448):
 * F is the (n_files, n_crops, D) embedding block, ia / ib index files, out[p] is
 *   SVHIP_TRIAL_COSINE : mean_i | cos(F[a, i], F[b, i]) |, per-norm clamp 1e-5              (utils.py:163-164)
 *   SVHIP_TRIAL_PNORM  : mean_i || F[a, i] - F[b, i] + 1e-6 ||_2                            (utils.py:167-169)
 *   SVHIP_TRIAL_PDIST  : - mean_{i,d} sqrt( sum_j (F[a, i, d] - F[b, j, d] + 1e-6)^2 )      (model.py:425-431, cohorts_path = None)
 * mean_crops: out (n_files, D) = mean over the crops (the AS-norm statement runs on crop means).
 * Pointers follow `flags`. */
enum { SVHIP_TRIAL_COSINE = 0, SVHIP_TRIAL_PNORM = 1, SVHIP_TRIAL_PDIST = 2 };
int svhip_score_trials(svhip_handle* h, int32_t mode, const float* F, int64_t n_files, int32_t n_crops, int32_t D,
                       const int32_t* ia, const int32_t* ib, int64_t P, float* out, int32_t flags);
int svhip_mean_crops(svhip_handle* h, const float* F, int64_t n_files, int32_t n_crops, int32_t D, float* out, int32_t flags);
/* SVHIP_TRIAL_PNORM with the reference's `p` argument (pnorm_similarity(ref, com, p), utils.py:167-169 -> F.pairwise_distance(p=p, eps=1e-6)):
 * out[t] = mean_i || F[a, i] - F[b, i] + 1e-6 ||_p; p = +-inf: max / min |d|, p = 0: the count of non-zero d, otherwise (sum |d|^p)^(1/p).
 * p = 2 is svhip_score_trials(SVHIP_TRIAL_PNORM). */
int svhip_score_trials_pnorm(svhip_handle* h, float p, const float* F, int64_t n_files, int32_t n_crops, int32_t D,
                             const int32_t* ia, const int32_t* ib, int64_t P, float* out, int32_t flags);

/* Verification metrics over a scored trial list: the sort-and-accumulate core of the reference's
 * evaluation tail, which it runs as Python list sorts and loops (src/utils.py:221-275) and sklearn calls (utils.py:74-121).
 * scores fp32, labels int32 in {0, 1}, P trials (< 2^31); pointers follow `flags`, the scalar results are host pointers.
 *   roc_points  : sklearn's _binary_clf_curve as roc_curve / precision_recall_curve use it in tuneThresholdfromScore
 *                 (utils.py:77-80,112): scores through nan_to_num, then one point per DISTINCT score value, highest first:
 *                 thr[k], fps[k] / tps[k] = negatives / positives scoring >= thr[k].  *n_out distinct values; the output
 *                 buffers hold P entries.  The caller finishes with the reference's own O(n) selection (drop_intermediate,
 *                 argmin / argmax, trapezoid) — speakerverification_amd/metrics.py does, in float64 like the reference.
 *   error_rates : ComputeErrorRates (utils.py:221-256): stable ascending sort (ties keep list order, as sorted() does);
 *                 fnrs[i] = positives among the i+1 lowest / positives, fprs[i] = 1 - negatives among them / negatives
 *                 (float64), thresholds[i] = the i-th lowest score.
 *   min_dcf     : ComputeErrorRates + ComputeMinDcf (utils.py:262-275) fused: the FIRST minimum over i of
 *                 c_miss*fnrs[i]*p_target + c_fa*fprs[i]*(1-p_target), divided by min(c_miss*p_target, c_fa*(1-p_target));
 *                 float64, Python's operation order, no fused multiply-add. */
int svhip_roc_points(svhip_handle* h, const float* scores, const int32_t* labels, int64_t P, int64_t* n_out, float* thr,
                     int64_t* fps, int64_t* tps, int32_t flags);
int svhip_error_rates(svhip_handle* h, const float* scores, const int32_t* labels, int64_t P, double* fnrs, double* fprs,
                      float* thresholds, int32_t flags);
int svhip_min_dcf(svhip_handle* h, const float* scores, const int32_t* labels, int64_t P, double p_target, double c_miss,
                  double c_fa, double* min_dcf, float* threshold, int32_t flags);

/* Packed checkpoint blob.  Replaces, at deployment time, the Python-side checkpoint handling of
 * ModelHandling.loadParameters (src/model.py:718-746: torch.load of a '.model' file + name-matched copy, and
 * src/trainer.py:145-205 which writes those files): speakerverification_amd/checkpoint.py converts a reference state dict
 * once (host-side torch.load) into a flat, versioned, FNV-1a-checksummed tensor archive; svhip_load_blob mmaps it and feeds
 * every tensor through svhip_load_tensor + svhip_finalize_weights (BatchNorm fold, weight packing, sinc filter bake), so no
 * Python / torch is needed at run time and the result is bit-identical to the state-dict path.
 *   blob_write  : n tensors (names are reference state_dict keys of __S__), shapes n x 4 (unused dims ignored), dtypes SVHIP_*.
 *   blob_open   : mmap + validate (magic, version, size, checksum, every table entry in bounds).  Errors: svhip_blob_last_error().
 *   blob_tensor : borrow tensor idx (pointers stay valid until blob_close).
 *   load_blob   : tensors whose names the handle's model does not know are skipped (as loadParameters skips them,
 *                 model.py:730-736); a known name with the wrong shape, a missing tensor, or a model mismatch is an error. */
typedef struct svhip_blob svhip_blob;
int svhip_blob_write(const char* path, int32_t model, int32_t n, const char* const* names, const void* const* data,
                     const int64_t* shapes, const int32_t* ndims, const int32_t* dtypes);
int svhip_blob_open(const char* path, svhip_blob** out);
int32_t svhip_blob_count(const svhip_blob* b);
int32_t svhip_blob_model(const svhip_blob* b);
int svhip_blob_tensor(const svhip_blob* b, int32_t idx, const char** name, const void** data, int64_t* shape4,
                      int32_t* ndim, int32_t* dtype);
int svhip_blob_close(svhip_blob* b);
const char* svhip_blob_last_error(void);
int svhip_load_blob(svhip_handle* h, const char* path);

/* Multi-GPU exchange.  Replaces the reference's `torch.distributed.all_gather_object` of per-rank
 * feature dicts (src/model.py:400-411): one process per GPU, utterances sharded in contiguous blocks, ONE RCCL all-gather
 * of the dense (rows, D) fp32 block per rank on the handle's stream — embed -> gather -> score stays on the device.
 *   comm_unique_id : rank 0 draws the 128-byte RCCL id; the host ships it to the other ranks by any side channel
 *                    (speakerverification_amd/distributed.py: the torch.distributed store).  Errors: svhip_comm_last_error().
 *   comm_init      : ncclCommInitRank on the handle's device (collective: every rank of `world` must call it).
 *   allgather_rows : out (world * rows, D) <- every rank's local (rows, D); rank r's block lands at out + r * rows * D.
 *                    All ranks pass the same `rows` (pad the last block).  Pointers follow `flags`.
 *   comm_destroy   : also called by svhip_destroy.
 * RCCL is bound at the first svhip_comm_* call (dlopen librccl.so.1); without it these return SVHIP_ERR_UNSUPPORTED. */
#define SVHIP_COMM_ID_BYTES 128
int svhip_comm_unique_id(void* id_out);
int svhip_comm_init(svhip_handle* h, const void* id, int32_t rank, int32_t world);
int svhip_comm_rank(const svhip_handle* h, int32_t* rank, int32_t* world);
int svhip_allgather_rows(svhip_handle* h, const float* local, int64_t rows, int32_t D, float* out, int32_t flags);
int svhip_comm_destroy(svhip_handle* h);
const char* svhip_comm_last_error(void);

/* Synthetic workload generator (nothing in the reference corresponds — its evaluation reads files):
 * utterances [first_utt, first_utt + B) of the counter-based stream `seed` (Philox4x32-10 + Box-Muller, 0.1 * N(0,1)
 * clipped to [-1, 1]), (B, L) fp32, L % 4 == 0.  A pure function of (seed, utterance, sample): every rank of a sharded
 * run generates exactly its own block without moving waveforms over PCIe.  oracle/synthwave.py restates it. */
int svhip_synth_waveforms(svhip_handle* h, uint64_t seed, int64_t first_utt, int32_t B, int32_t L, float* wav_out, int32_t flags);

/* Introspection used by tests and bench.py (not part of the reference's surface).
 *   get_stage    : copy an intermediate activation of the LAST forward to host as fp32, frame-major
 *                  (B, T, C).  Names: "input","blocks.0".."blocks.3","mfa","asp","asp_bn" (ECAPA).
 *                  Returns the element count through *count (out may be NULL to query).
 *   profile_*    : when enabled every kernel launch is bracketed by HIP events on the handle's
 *                  stream; profile_get returns accumulated milliseconds / launch count per kernel
 *                  label since the last reset (enumerate idx = 0.. until SVHIP_ERR_INVALID).
 *   workload_flops: algorithmic FLOPs (2 x MACs of conv/linear layers) of one utterance. */
int svhip_get_stage(svhip_handle* h, const char* name, float* out, int64_t* count);
int svhip_profile_enable(svhip_handle* h, int32_t on);
int svhip_profile_filter(svhip_handle* h, const char* label);   /* NULL / "": every kernel; else only this label (fewer events in the stream) */
int svhip_profile_reset(svhip_handle* h);
int svhip_profile_get(svhip_handle* h, int32_t idx, char* name, int32_t name_cap, double* ms, int64_t* launches, double* flops);
double svhip_workload_flops(const svhip_handle* h);
/* Developer / test options (not part of the reference's surface).  The SVHIP_<NAME> environment variables are read ONCE, by
 * svhip_create, as a new handle's defaults; afterwards only this call changes them — no getenv on the hot path.  Names:
 * "pw3_cus" (cap of the persistent GEMM grids; 0: off), "rn_unfused", "rn_stop", "rn_snap", "asp_v1", "r2_big", "x3_keep_f32",
 * "asnorm_slab", "asnorm_f32mfma", "asnorm_x6", "asnorm_2s", "asnorm_w32", "score_f32mfma", "score_tiled", "fbank32", "cv_off", "pw3_tail_off", "n128_off", "r2_slices", "rn_tail_big", "rn_sinc_f32", "rn_step_off", "rn_pool_off", "layer_labels".  Unknown names: SVHIP_ERR_INVALID. */
int svhip_set_option(svhip_handle* h, const char* name, int32_t value);
/* Free the scoring / metrics scratch slots of the handle (grown on demand, otherwise kept until svhip_destroy). */
int svhip_trim_scratch(svhip_handle* h);
int svhip_selftest(void);   /* host-only self checks (per-device launch-attribute bookkeeping); 0 = ok, no GPU needed */

#ifdef __cplusplus
}
#endif
#endif /* SVHIP_H */
