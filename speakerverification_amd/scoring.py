"""Scoring counterparts of the reference's ``src/utils.py:126-169``.

``similarity_measure(method, ref, com, **kw)`` keeps the per-trial signature for compatibility; the
batched entry points (``score_pairs``, ``score_matrix``, ``asnorm_stats``, ``asnorm_pairs``,
``score_trials``) are what ``ModelHandling.evaluateFromList`` uses: one kernel launch per trial LIST
instead of three host<->device crossings per trial.
"""
from __future__ import annotations

import numpy as np

from .engine import Engine, _is_torch

try:
    import torch
except Exception:  # pragma: no cover
    torch = None

_engines = {}


def scoring_engine(device=0) -> Engine:
    """one fbank/scoring-only handle per device, created on first use"""
    if device not in _engines:
        _engines[device] = Engine(model="none", max_batch=1, device=device)
    return _engines[device]


_comms = {}


def lib_comm(device, rank, world):
    """the RCCL communicator of a device's scoring engine (svhip_comm_*): created once per process and device — svhip_comm_init
    refuses a second communicator on a handle — and reused by every ModelHandling; destroyed with the engine."""
    from . import distributed as sv_dist
    key = (device, rank, world)
    if key not in _comms:
        _comms[key] = sv_dist.LibComm(scoring_engine(device), rank, world)
    return _comms[key]


def _np(x):
    return x.detach().cpu().numpy() if _is_torch(x) else np.asarray(x)


# ---- batched API ---------------------------------------------------------------------------------------
def score_pairs(E, ia, ib, device=0):
    """|cos(E[ia], E[ib])| (utils.py:163-164 with one crop per row)."""
    return scoring_engine(device).score_pairs(E, ia, ib)


def score_matrix(A, B, device=0):
    return scoring_engine(device).score_matrix(A, B)


def asnorm_stats(E, cohort, top=200, device=0):
    return scoring_engine(device).asnorm_stats(E, cohort, top)


def asnorm_pairs(E, mu, sigma, ia, ib, device=0):
    return scoring_engine(device).asnorm_pairs(E, mu, sigma, ia, ib)


def score_trials(feats, ia, ib, mode="cosine", cohorts=None, top=200, device=0):
    """Score a whole trial list.  ``feats``: (n_files, n_crops, D) float32 (already L2-normalised when
    the loss sets test_normalize) — a numpy array, or a CUDA tensor, in which case everything stays on the device and a CUDA
    tensor of P scores comes back; ``ia``/``ib``: file indices per trial.
      cosine: mean_i |cos(R_i, C_i)| over aligned crops                          (utils.py:163-164)
      norm  : adaptive S-norm on crop means with the top-`top` cohort scores     (utils.py:135-160)
      pnorm : mean_i ||R_i - C_i + 1e-6||_2                                      (utils.py:167-169)
      pdist : -mean pairwise distance over the (n, D, n) broadcast               (model.py:425-431, cohorts_path=None)
    Every mode is a device kernel (svhip_score_trials / svhip_mean_crops + svhip_asnorm_*)."""
    from . import engine as _engine
    on_dev = _is_torch(feats) and feats.is_cuda
    if not on_dev:
        feats = np.ascontiguousarray(_np(feats), dtype=np.float32)
    if feats.ndim == 2:
        feats = feats[:, None, :]
    ia = np.ascontiguousarray(ia, dtype=np.int32)
    ib = np.ascontiguousarray(ib, dtype=np.int32)
    eng = scoring_engine(device)
    if len(ia) == 0:
        if on_dev:
            import torch
            return torch.zeros((0,), dtype=torch.float32, device=feats.device)
        return np.zeros((0,), np.float32)
    # the C ABI range-checks host indices only; on the device-resident path they are uploaded first, so the check happens here (O(P))
    n_files = int(feats.shape[0])
    if int(ia.min()) < 0 or int(ib.min()) < 0 or int(ia.max()) >= n_files or int(ib.max()) >= n_files:
        raise ValueError(f"trial indexes outside [0, {n_files})")
    if on_dev:                                # indices follow the embeddings into HBM (8 P bytes over PCIe)
        feats = feats.contiguous()
        ia, ib = _engine.to_device(ia, feats.device.index), _engine.to_device(ib, feats.device.index)
    if mode in ("cosine", "pnorm", "pdist"):
        return eng.score_trials(feats, ia, ib, mode)
    if mode in ("norm", "zt_norm"):
        if cohorts is None:
            raise ValueError("scoring_mode 'norm' needs a cohort matrix")
        Em = eng.mean_crops(feats)                                              # crop means
        if on_dev:
            cohorts = cohorts if (_is_torch(cohorts) and cohorts.is_cuda) else _engine.to_device(np.ascontiguousarray(_np(cohorts), np.float32), feats.device.index)
        else:
            cohorts = np.ascontiguousarray(_np(cohorts), dtype=np.float32)
        mu, sd = eng.asnorm_stats(Em, cohorts, top)
        return eng.asnorm_pairs(Em, mu, sd, ia, ib)
    raise ValueError(f"unknown scoring mode {mode}")


# ---- per-trial compatibility API (same names / arguments as the reference) ---------------------------------
def cosine_similarity(ref, com, **kwargs):
    r, c = np.ascontiguousarray(_np(ref), np.float32), np.ascontiguousarray(_np(com), np.float32)
    if r.ndim == 1:
        r, c = r[None], c[None]
    n = r.shape[0]
    E = np.concatenate([r, c])
    s = scoring_engine(kwargs.get("device", 0)).score_pairs(E, np.arange(n, dtype=np.int32), np.arange(n, 2 * n, dtype=np.int32))
    return np.mean(s)


def ZT_norm_similarity(ref, com, cohorts, top=-1, **kwargs):
    r, c = _np(ref).astype(np.float32), _np(com).astype(np.float32)
    if r.ndim == 1:
        r, c = r[None], c[None]
    E = np.ascontiguousarray(np.stack([r.mean(axis=0), c.mean(axis=0)]), dtype=np.float32)
    eng = scoring_engine(kwargs.get("device", 0))
    mu, sd = eng.asnorm_stats(E, np.ascontiguousarray(_np(cohorts), np.float32), top)
    return float(eng.asnorm_pairs(E, mu, sd, np.array([0], np.int32), np.array([1], np.int32))[0])


def pnorm_similarity(ref, com, p=2, **kwargs):
    r, c = np.ascontiguousarray(_np(ref), np.float32), np.ascontiguousarray(_np(com), np.float32)
    if r.ndim == 1:
        r, c = r[None], c[None]
    F = np.ascontiguousarray(np.stack([r, c]), dtype=np.float32)                 # (2 files, n crops, D)
    eng = scoring_engine(kwargs.get("device", 0))
    # (the reference only ever calls it with p = 2, model.py:446; any p of F.pairwise_distance is served: svhip_score_trials_pnorm)
    return float(eng.score_trials(F, np.array([0], np.int32), np.array([1], np.int32), "pnorm", p=p)[0])


def similarity_measure(method="cosine", ref=None, com=None, **kwargs):
    """src/utils.py:126-132"""
    if method == "cosine":
        return cosine_similarity(ref, com, **kwargs)
    elif method == "pnorm":
        return pnorm_similarity(ref, com, **kwargs)
    elif method == "zt_norm":
        return ZT_norm_similarity(ref, com, **kwargs)
