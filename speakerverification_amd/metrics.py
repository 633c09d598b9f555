"""Verification metrics with the reference's names and return layouts (src/utils.py:74-121, 221-275), computed from the
device-side sort / scan (csrc/metrics.hip) instead of Python list sorts and sklearn calls.

The O(P log P) work — sorting the trial scores and accumulating the labels — runs on the GPU; what is left is the
reference's own O(n) selection arithmetic on the resulting curve (sklearn's drop_intermediate rule, argmin / argmax,
trapezoid AUC), done here in float64 numpy exactly as the reference does it, so the numbers agree to the last bit for
float32 scores.  (Scores that are not exactly representable in float32 are rounded on entry; the reference's scores come
out of float32 tensors, so they are.)
"""
from __future__ import annotations

import numpy as np

from .engine import Engine

_engine = None


def _default_engine():
    global _engine
    if _engine is None:
        _engine = Engine(model="none", max_batch=1)
    return _engine


def _roc_curve(fps, tps, thr):
    """sklearn.metrics.roc_curve (1.3+) after _binary_clf_curve: drop collinear points, prepend (0, 0) at threshold inf."""
    if len(fps) > 2:
        keep = np.where(np.r_[True, np.logical_or(np.diff(fps, 2), np.diff(tps, 2)), True])[0]
        fps, tps, thr = fps[keep], tps[keep], thr[keep]
    tps = np.r_[0, tps]
    fps = np.r_[0, fps]
    thr = np.r_[np.inf, thr.astype(np.float64)]
    fpr = np.repeat(np.nan, fps.shape) if fps[-1] <= 0 else fps / fps[-1]
    tpr = np.repeat(np.nan, tps.shape) if tps[-1] <= 0 else tps / tps[-1]
    return fpr, tpr, thr


def _precision_recall_curve(fps, tps, thr):
    """sklearn.metrics.precision_recall_curve (drop_intermediate=False) after _binary_clf_curve."""
    ps = tps + fps
    precision = np.zeros_like(tps)
    np.divide(tps, ps, out=precision, where=(ps != 0))
    recall = np.ones_like(tps) if tps[-1] == 0 else tps / tps[-1]
    sl = slice(None, None, -1)
    return np.hstack((precision[sl], 1)), np.hstack((recall[sl], 0)), thr.astype(np.float64)[sl]


def _auc(x, y):
    """sklearn.metrics.auc for a monotonic x: trapezoid, sign by direction."""
    dx = np.diff(x)
    direction = -1 if (np.any(dx < 0) and np.all(dx <= 0)) else 1
    if np.any(dx < 0) and not np.all(dx <= 0):
        raise ValueError("x is neither increasing nor decreasing")
    trap = getattr(np, "trapezoid", None) or np.trapz
    return float(direction * trap(y, x))


def tuneThresholdfromScore(scores, labels, target_fa, target_fr=None, engine=None):
    """src/utils.py:74-121: same dict ('gmean', 'roc', 'prec_recall'), same list layouts."""
    eng = engine or _default_engine()
    labels = np.nan_to_num(np.asarray(labels, dtype=np.float64))
    fps, tps, thr = eng.roc_points(scores, labels != 0)
    fpr, tpr, thresholds = _roc_curve(fps, tps, thr)
    results = {}
    gmean = np.sqrt(tpr * (1 - fpr))
    idxG = np.argmax(gmean)
    G_mean_result = [idxG, gmean[idxG], thresholds[idxG]]
    fnr = (1 - tpr) * 100
    fpr = fpr * 100
    tuned = []
    if target_fr:
        for tfr in target_fr:
            idx = np.nanargmin(np.absolute(tfr - fnr))
            tuned.append([thresholds[idx], fpr[idx], fnr[idx]])
    for tfa in target_fa:
        idx = np.nanargmin(np.absolute(tfa - fpr))
        tuned.append([thresholds[idx], fpr[idx], fnr[idx]])
    idxE = np.nanargmin(np.absolute(fnr - fpr))
    eer = np.mean([fpr[idxE], fnr[idxE]])
    optimal_threshold = thresholds[idxE]
    precision, recall, thresholds_ = _precision_recall_curve(fps, tps, thr)
    with np.errstate(invalid="ignore", divide="ignore"):
        fscore = (2 * precision * recall) / (precision + recall)
    ixPR = np.argmax(fscore)
    results["gmean"] = G_mean_result
    results["roc"] = [tuned, eer, _auc(fpr, tpr), optimal_threshold]
    results["prec_recall"] = [precision, recall, fscore[ixPR], thresholds_[ixPR]]
    return results


def ComputeErrorRates(scores, labels, engine=None):
    """src/utils.py:221-256 -> (fnrs, fprs, thresholds); arrays instead of Python lists, same values and order."""
    eng = engine or _default_engine()
    return eng.error_rates(scores, labels)


def ComputeMinDcf(fnrs, fprs, thresholds, p_target, c_miss, c_fa):
    """src/utils.py:262-275 on the arrays ComputeErrorRates returned (vectorised; first minimum, like the loop's '<')."""
    fnrs, fprs = np.asarray(fnrs, np.float64), np.asarray(fprs, np.float64)
    c_det = c_miss * fnrs * p_target + c_fa * fprs * (1 - p_target)
    i = int(np.argmin(c_det))
    c_def = min(c_miss * p_target, c_fa * (1 - p_target))
    return float(c_det[i] / c_def), thresholds[i]


def min_dcf(scores, labels, p_target, c_miss, c_fa, engine=None):
    """ComputeErrorRates + ComputeMinDcf in one device call."""
    eng = engine or _default_engine()
    return eng.min_dcf(scores, labels, p_target, c_miss, c_fa)
