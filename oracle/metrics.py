"""Oracle (TEST INFRASTRUCTURE) — verification metrics as the reference computes them, restated on numpy.

``src/utils.py:74-121`` (``tuneThresholdfromScore``: sklearn ``roc_curve`` / ``precision_recall_curve`` / ``auc``) and
``src/utils.py:221-275`` (``ComputeErrorRates`` / ``ComputeMinDcf``: a stable Python sort and two loops).  The sklearn
pieces are restated from scikit-learn 1.7 (``sklearn/metrics/_ranking.py``: ``_binary_clf_curve``, ``roc_curve`` with its
default ``drop_intermediate=True``, ``precision_recall_curve`` with its default ``drop_intermediate=False``, ``auc``).
PINNED by ``oracle/make_golden.py::golden_metrics`` against the imported reference (which calls the installed sklearn) ->
``tests/golden/metrics.npz``.  Nothing here is imported by the product.
"""
from __future__ import annotations

import numpy as np


def binary_clf_curve(labels, scores):
    """sklearn _binary_clf_curve: (fps, tps, thresholds), one point per distinct score, highest first (float64)."""
    y_true = np.asarray(labels) == 1
    y_score = np.asarray(scores, dtype=np.float64)
    order = np.argsort(y_score, kind="mergesort")[::-1]
    y_score, y_true = y_score[order], y_true[order]
    distinct = np.where(np.diff(y_score))[0]
    idx = np.r_[distinct, y_true.size - 1]
    tps = np.cumsum(y_true * 1.0, dtype=np.float64)[idx]
    fps = 1 + idx - tps
    return fps, tps, y_score[idx]


def roc_curve(labels, scores):
    fps, tps, thr = binary_clf_curve(labels, scores)
    if len(fps) > 2:
        keep = np.where(np.r_[True, np.logical_or(np.diff(fps, 2), np.diff(tps, 2)), True])[0]
        fps, tps, thr = fps[keep], tps[keep], thr[keep]
    tps, fps, thr = np.r_[0, tps], np.r_[0, fps], np.r_[np.inf, thr]
    return fps / fps[-1], tps / tps[-1], thr


def precision_recall_curve(labels, scores):
    fps, tps, thr = binary_clf_curve(labels, scores)
    ps = tps + fps
    precision = np.zeros_like(tps)
    np.divide(tps, ps, out=precision, where=(ps != 0))
    recall = tps / tps[-1]
    sl = slice(None, None, -1)
    return np.hstack((precision[sl], 1)), np.hstack((recall[sl], 0)), thr[sl]


def tune_threshold_from_score(scores, labels, target_fa, target_fr=None):
    """src/utils.py:74-121, line by line."""
    labels = np.nan_to_num(labels)
    scores = np.nan_to_num(scores)
    fpr, tpr, thresholds = roc_curve(labels, scores)
    gmean = np.sqrt(tpr * (1 - fpr))
    idxG = np.argmax(gmean)
    G = [idxG, gmean[idxG], thresholds[idxG]]
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
    precision, recall, thr_ = precision_recall_curve(labels, scores)
    with np.errstate(invalid="ignore", divide="ignore"):
        fscore = (2 * precision * recall) / (precision + recall)
    ix = np.argmax(fscore)
    trap = getattr(np, "trapezoid", None) or np.trapz
    return {"gmean": G, "roc": [tuned, eer, float(trap(tpr, fpr)), thresholds[idxE]],
            "prec_recall": [precision, recall, fscore[ix], thr_[ix]]}


def compute_error_rates(scores, labels):
    """src/utils.py:221-256 (stable ascending sort; float64 ratios)."""
    scores = np.asarray(scores, dtype=np.float64)
    labels = np.asarray(labels, dtype=np.int64)
    order = np.argsort(scores, kind="stable")
    lab = labels[order]
    cpos = np.cumsum(lab)
    cneg = np.cumsum(1 - lab)
    fnrs = cpos / float(cpos[-1])
    fprs = 1 - cneg / float(len(lab) - cpos[-1])
    return fnrs, fprs, scores[order]


def compute_min_dcf(fnrs, fprs, thresholds, p_target, c_miss, c_fa):
    """src/utils.py:262-275: first minimum of the detection cost, normalised."""
    c_det = c_miss * np.asarray(fnrs) * p_target + c_fa * np.asarray(fprs) * (1 - p_target)
    i = int(np.argmin(c_det))
    return float(c_det[i] / min(c_miss * p_target, c_fa * (1 - p_target))), thresholds[i]
