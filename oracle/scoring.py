"""Oracle (TEST INFRASTRUCTURE) — trial scoring and eval-mode cropping, as the reference does them.

Per-trial forms follow ``src/utils.py:126-169``; the batched forms are the algebraically equal
GEMM statements used to check the HIP scorers at sizes where a per-trial
Python loop is too slow.  PINNED by ``oracle/make_golden.py`` -> ``tests/golden/scoring.npz``.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def l2_normalize(e: torch.Tensor) -> torch.Tensor:
    """src/model.py:421-423: F.normalize(p=2, dim=1) (eps=1e-12 on the norm)."""
    return F.normalize(e, p=2, dim=1)


def cosine_similarity(ref: torch.Tensor, com: torch.Tensor) -> float:
    """src/utils.py:163-164: mean_i | cos(ref_i, com_i) |, crop-aligned, eps=1e-5."""
    return float(np.mean(abs(F.cosine_similarity(ref, com, dim=-1, eps=1e-05)).cpu().numpy()))


def pnorm_similarity(ref: torch.Tensor, com: torch.Tensor, p=2) -> float:
    """src/utils.py:167-169: mean pairwise p-distance with eps=1e-6 added to the difference."""
    return float(np.mean(F.pairwise_distance(ref, com, p=p, eps=1e-06, keepdim=True).numpy()))


def zt_norm_similarity(ref, com, cohorts, top=-1) -> float:
    """src/utils.py:135-160 adaptive symmetric score normalisation (per trial).
    NB ``[:top]`` with the default top=-1 drops the smallest cohort score — kept as is."""
    ref = np.asarray(ref, dtype=np.float32)
    com = np.asarray(com, dtype=np.float32)

    def zt(a, b):
        S = np.mean(np.inner(cohorts, a), axis=1)
        S = np.sort(S, axis=0)[::-1][:top]
        return (np.mean(np.inner(a, b)) - np.mean(S)) / np.std(S)

    return float((zt(ref, com) + zt(com, ref)) / 2)


# ---- batched statements (one crop per utterance unless stated) -------------------------------

def cosine_pairs(E: np.ndarray, ia: np.ndarray, ib: np.ndarray, eps=1e-5) -> np.ndarray:
    """|cos(E[ia], E[ib])| for a pair list; float64 accumulation, per-norm clamp at eps
    (torch>=1.12 semantics of F.cosine_similarity, identical to the older product clamp for
    unit-norm inputs)."""
    a = E[ia].astype(np.float64)
    b = E[ib].astype(np.float64)
    na = np.maximum(np.linalg.norm(a, axis=1), eps)
    nb = np.maximum(np.linalg.norm(b, axis=1), eps)
    return np.abs(np.sum(a * b, axis=1) / (na * nb))


def asnorm_stats(E: np.ndarray, cohort: np.ndarray, top: int):
    """Per-embedding cohort statistics of utils.py:142-146 on crop means: S = cohort @ e,
    sort descending, keep [:top], population mean/std.  float64.  Returns (mu[N], sigma[N])."""
    S = E.astype(np.float64) @ cohort.astype(np.float64).T
    S = -np.sort(-S, axis=1)[:, :top]
    return S.mean(axis=1), S.std(axis=1)


def asnorm_pairs(E, ia, ib, cohort, top):
    """0.5*((s-mu_a)/sd_a + (s-mu_b)/sd_b), s = E[ia]·E[ib] (== utils.py:155-160 for one crop)."""
    mu, sd = asnorm_stats(E, cohort, top)
    s = np.sum(E[ia].astype(np.float64) * E[ib].astype(np.float64), axis=1)
    return 0.5 * ((s - mu[ia]) / sd[ia] + (s - mu[ib]) / sd[ib])


def crop_eval(audio: np.ndarray, max_audio=32000, num_eval=10, peak_normalize=True) -> np.ndarray:
    """src/processing/audio_loader.py:100-150 for an ndarray source in eval mode:
    peak normalise (wav_conversion.py:35-41), wrap-pad short audio to max_audio+1, take num_eval
    crops at linspace(0, len-max_audio, num_eval) (int() truncation)."""
    if peak_normalize:
        if np.issubdtype(audio.dtype, np.integer):
            info = np.iinfo(audio.dtype)
            audio = audio / max(info.max, -info.min)
        else:
            audio = audio / max(audio.max(), -audio.min())
    n = audio.shape[0]
    if n <= max_audio:
        audio = np.pad(audio, (0, max_audio - n + 1), "wrap")
        n = audio.shape[0]
    if num_eval == 0:
        return np.stack([audio], axis=0).astype(np.float32)
    starts = np.linspace(0, n - max_audio, num=num_eval)
    return np.stack([audio[int(s):int(s) + max_audio] for s in starts], axis=0).astype(np.float32)
