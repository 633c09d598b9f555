"""ctypes binding of libsvhip.so (the C ABI declared in include/svhip.h).

There is no CPU fallback: if the HIP library is missing or fails to load the import of any
product module that needs it raises ``SvhipUnavailable`` — loudly — instead of silently running
something else.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SVHIP_LIB_PATH") or os.path.join(HERE, "libsvhip.so")     # (override: developer A/B of two builds)

OK = 0
ERR_NONFINITE, ERR_RANGE = -7, -8          # the call completed and wrote its outputs; they are numerically suspect (include/svhip.h)
MODEL_ECAPA, MODEL_RAWNET2, MODEL_NONE = 0, 1, 2
F32, BF16, I64, F32X3, F16 = 0, 1, 2, 3, 4
IN_DEVICE, OUT_DEVICE, ASYNC = 1, 2, 4
TRIAL_COSINE, TRIAL_PNORM, TRIAL_PDIST = 0, 1, 2
COMM_ID_BYTES = 128


class SvhipUnavailable(RuntimeError):
    pass


class SvhipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"svhip error {code}: {msg}")
        self.code = code


class SvhipNumericError(SvhipError):
    """SVHIP_ERR_NONFINITE / SVHIP_ERR_RANGE: the call completed and wrote its outputs, but they are numerically suspect (an fp16
    activation overflowed, the input left the range of an f32x3 handle's half-precision planes, the input was not finite)."""


class Config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("model", C.c_int32), ("compute", C.c_int32), ("device", C.c_int32),
        ("channels", C.c_int32), ("n_mels", C.c_int32), ("embed_dim", C.c_int32), ("max_batch", C.c_int32),
        ("samples", C.c_int32), ("log_input", C.c_int32), ("input_norm", C.c_int32),
        ("fb_sr", C.c_int32), ("n_fft", C.c_int32), ("win_length", C.c_int32), ("hop_length", C.c_int32),
        ("fmin", C.c_float), ("fmax", C.c_float), ("preemph", C.c_float),
        ("stream", C.c_void_p),
    ]


# every symbol include/svhip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_SIGNATURES = {
    "svhip_default_config": (None, [C.POINTER(Config)]),
    "svhip_abi_version": (C.c_int, []),
    "svhip_create": (C.c_int, [C.POINTER(Config), C.POINTER(_P)]),
    "svhip_destroy": (C.c_int, [_P]),
    "svhip_last_error": (C.c_char_p, [_P]),
    "svhip_synchronize": (C.c_int, [_P]),
    "svhip_numeric_status": (C.c_int, [_P, C.c_int32]),
    "svhip_load_tensor": (C.c_int, [_P, C.c_char_p, _P, C.POINTER(C.c_int64), C.c_int32, C.c_int32]),
    "svhip_finalize_weights": (C.c_int, [_P]),
    "svhip_fbank": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, C.c_int32]),
    "svhip_embed_features": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, C.c_int32]),
    "svhip_embed_wave": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, C.c_int32]),
    "svhip_crop_pcm16": (C.c_int, [_P, _P, C.c_int64, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32]),
    "svhip_l2norm": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int32]),
    "svhip_score_pairs": (C.c_int, [_P, _P, C.c_int64, C.c_int32, _P, _P, C.c_int64, _P, C.c_int32]),
    "svhip_score_matrix": (C.c_int, [_P, _P, C.c_int64, _P, C.c_int64, C.c_int32, _P, C.c_int32]),
    "svhip_asnorm_stats": (C.c_int, [_P, _P, C.c_int64, C.c_int32, _P, C.c_int32, C.c_int32, _P, _P, C.c_int32]),
    "svhip_asnorm_pairs": (C.c_int, [_P, _P, C.c_int64, C.c_int32, _P, _P, _P, _P, C.c_int64, _P, C.c_int32]),
    "svhip_asnorm_last_fallback": (C.c_int64, [_P]),
    "svhip_asnorm_last_refit": (C.c_int64, [_P, C.POINTER(C.c_int32)]),
    "svhip_score_trials": (C.c_int, [_P, C.c_int32, _P, C.c_int64, C.c_int32, C.c_int32, _P, _P, C.c_int64, _P, C.c_int32]),
    "svhip_mean_crops": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int32, _P, C.c_int32]),
    "svhip_score_trials_pnorm": (C.c_int, [_P, C.c_float, _P, C.c_int64, C.c_int32, C.c_int32, _P, _P, C.c_int64, _P, C.c_int32]),
    "svhip_roc_points": (C.c_int, [_P, _P, _P, C.c_int64, C.POINTER(C.c_int64), _P, _P, _P, C.c_int32]),
    "svhip_error_rates": (C.c_int, [_P, _P, _P, C.c_int64, _P, _P, _P, C.c_int32]),
    "svhip_min_dcf": (C.c_int, [_P, _P, _P, C.c_int64, C.c_double, C.c_double, C.c_double, C.POINTER(C.c_double),
                                C.POINTER(C.c_float), C.c_int32]),
    "svhip_blob_write": (C.c_int, [C.c_char_p, C.c_int32, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(_P), C.POINTER(C.c_int64),
                                   C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "svhip_blob_open": (C.c_int, [C.c_char_p, C.POINTER(_P)]),
    "svhip_blob_count": (C.c_int32, [_P]),
    "svhip_blob_model": (C.c_int32, [_P]),
    "svhip_blob_tensor": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(_P), C.POINTER(C.c_int64),
                                    C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "svhip_blob_close": (C.c_int, [_P]),
    "svhip_blob_last_error": (C.c_char_p, []),
    "svhip_load_blob": (C.c_int, [_P, C.c_char_p]),
    "svhip_comm_unique_id": (C.c_int, [_P]),
    "svhip_comm_init": (C.c_int, [_P, _P, C.c_int32, C.c_int32]),
    "svhip_comm_rank": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "svhip_allgather_rows": (C.c_int, [_P, _P, C.c_int64, C.c_int32, _P, C.c_int32]),
    "svhip_comm_destroy": (C.c_int, [_P]),
    "svhip_comm_last_error": (C.c_char_p, []),
    "svhip_synth_waveforms": (C.c_int, [_P, C.c_uint64, C.c_int64, C.c_int32, C.c_int32, _P, C.c_int32]),
    "svhip_get_stage": (C.c_int, [_P, C.c_char_p, _P, C.POINTER(C.c_int64)]),
    "svhip_profile_enable": (C.c_int, [_P, C.c_int32]),
    "svhip_profile_filter": (C.c_int, [_P, C.c_char_p]),
    "svhip_profile_reset": (C.c_int, [_P]),
    "svhip_profile_get": (C.c_int, [_P, C.c_int32, C.c_char_p, C.c_int32, C.POINTER(C.c_double),
                                    C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "svhip_workload_flops": (C.c_double, [_P]),
    "svhip_set_option": (C.c_int, [_P, C.c_char_p, C.c_int32]),
    "svhip_trim_scratch": (C.c_int, [_P]),
    "svhip_selftest": (C.c_int, []),
}

_lib = None


def exported_symbols():
    return sorted(_SIGNATURES)


def load():
    """Load libsvhip.so (once).  Raises SvhipUnavailable if it is not built or cannot be loaded."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SvhipUnavailable(
            f"{LIB_PATH} is missing: build it with `python -m speakerverification_amd.build` "
            "(hipcc, gfx950).  There is no CPU fallback for the product path.")
    # torch (plumbing: device memory, streams, torch.distributed) bundles its own HIP runtime; when it is
    # installed it must be the one runtime of the process, so load it before libsvhip resolves libamdhip64.
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - numpy-only use
        pass
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise SvhipUnavailable(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise SvhipUnavailable(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def default_config() -> Config:
    cfg = Config()
    load().svhip_default_config(C.byref(cfg))
    return cfg


def check(handle, rc, on_numeric="raise"):
    """rc of a C call -> exception.  The two numeric codes (the call DID complete and wrote its outputs) follow `on_numeric`:
    "raise" (SvhipNumericError), "warn" (RuntimeWarning, the caller gets the outputs as computed - what the reference, which never
    looks, would hand back) or "ignore"."""
    if rc == OK:
        return
    msg = load().svhip_last_error(handle)
    text = msg.decode() if msg else "?"
    if rc in (ERR_NONFINITE, ERR_RANGE):
        if on_numeric == "ignore":
            return
        if on_numeric == "warn":
            import warnings
            # (shown for EVERY batch that reports it: the default filter would print a call site's warning once and let later bad
            #  batches pass silently — ADVICE r5)
            with warnings.catch_warnings():
                warnings.simplefilter("always", RuntimeWarning)
                warnings.warn(f"svhip numeric status {rc}: {text}", RuntimeWarning, stacklevel=3)
            return
        raise SvhipNumericError(rc, text)
    raise SvhipError(rc, text)
