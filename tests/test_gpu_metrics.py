"""GPU: verification metrics (svhip_roc_points / svhip_error_rates / svhip_min_dcf + speakerverification_amd.metrics)
against the REFERENCE'S OWN tuneThresholdfromScore / ComputeErrorRates / ComputeMinDcf outputs (tests/golden/metrics.npz,
oracle/make_golden.py::golden_metrics) — bit for bit — and against the numpy oracle at the BASELINE trial-list size."""
import os

import numpy as np
import pytest

from oracle import metrics as o_metrics
from tests.metrics_data import metrics_case
from speakerverification_amd import _lib, metrics
from speakerverification_amd.engine import Engine

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = Engine(model="none", max_batch=1)
    yield e
    e.close()


@pytest.mark.parametrize("case", ["small", "ties", "distinct", "skewed"])
def test_matches_reference_outputs_exactly(eng, golden_dir, case):
    g = np.load(os.path.join(golden_dir, "metrics.npz"))
    sc, lab = metrics_case(case)
    res = metrics.tuneThresholdfromScore(sc, lab, [1, 0.1], [5], engine=eng)
    gm = g[case + "_gmean"]
    assert (int(res["gmean"][0]), float(res["gmean"][1]), float(res["gmean"][2])) == (int(gm[0]), gm[1], gm[2])
    assert np.array_equal(np.array(res["roc"][0], np.float64), g[case + "_tuned"])
    assert np.array_equal(np.array([res["roc"][1], res["roc"][2], res["roc"][3]], np.float64), g[case + "_eer_auc_thr"])
    assert np.array_equal(np.asarray(res["prec_recall"][0]), g[case + "_precision"])
    assert np.array_equal(np.asarray(res["prec_recall"][1]), g[case + "_recall"])
    assert np.array_equal(np.array([res["prec_recall"][2], res["prec_recall"][3]], np.float64), g[case + "_pr_best"])
    fnrs, fprs, thr = metrics.ComputeErrorRates(sc, lab, engine=eng)
    assert np.array_equal(fnrs, g[case + "_fnrs"]) and np.array_equal(fprs, g[case + "_fprs"])
    assert np.array_equal(thr.astype(np.float64), g[case + "_thr"])
    d = g[case + "_mindcf"]
    assert metrics.ComputeMinDcf(fnrs, fprs, thr, 0.05, 1, 1) == (d[0], np.float32(d[1]))
    assert metrics.ComputeMinDcf(fnrs, fprs, thr, 0.01, 10, 1) == (d[2], np.float32(d[3]))
    assert metrics.min_dcf(sc, lab, 0.05, 1, 1, engine=eng) == (d[0], float(np.float32(d[1])))     # fused device path
    assert metrics.min_dcf(sc, lab, 0.01, 10, 1, engine=eng) == (d[2], float(np.float32(d[3])))


def test_baseline_sized_trial_list_against_the_oracle(eng):
    """1.2 M trials (BASELINE config 4): every array the device returns equals the numpy restatement."""
    P = 1_200_000
    rng = np.random.Generator(np.random.PCG64(4))
    lab = (rng.random(P) < 0.5).astype(np.int64)
    sc = np.clip(0.25 * rng.standard_normal(P) + 0.35 * lab, -1, 1).astype(np.float32)
    sc[::7] = np.round(sc[::7] * 64) / 64                       # runs of tied scores, interleaved with distinct ones
    fps, tps, thr = eng.roc_points(sc, lab)
    ofps, otps, othr = o_metrics.binary_clf_curve(lab, sc)
    assert len(fps) == len(ofps) and np.array_equal(fps, ofps) and np.array_equal(tps, otps)
    assert np.array_equal(thr.astype(np.float64), othr)
    assert np.all(np.diff(thr) < 0) and fps[-1] + tps[-1] == P     # strictly descending thresholds, everything counted
    fnrs, fprs, t = eng.error_rates(sc, lab)
    ofn, ofp, ot = o_metrics.compute_error_rates(sc, lab)
    assert np.array_equal(fnrs, ofn) and np.array_equal(fprs, ofp) and np.array_equal(t.astype(np.float64), ot)
    for pt, cm, cf in ((0.05, 1, 1), (0.01, 10, 1), (0.5, 1, 3)):
        want = o_metrics.compute_min_dcf(ofn, ofp, ot, pt, cm, cf)
        assert eng.min_dcf(sc, lab, pt, cm, cf) == (want[0], float(want[1]))
    res = metrics.tuneThresholdfromScore(sc, lab, [1, 0.1], engine=eng)
    ores = o_metrics.tune_threshold_from_score(sc, lab, [1, 0.1])
    assert res["roc"][1] == ores["roc"][1] and res["roc"][2] == ores["roc"][2] and res["roc"][3] == ores["roc"][3]
    assert res["gmean"][1] == ores["gmean"][1] and np.array_equal(np.array(res["roc"][0]), np.array(ores["roc"][0]))


def test_nan_scores_and_edge_inputs(eng):
    sc = np.array([0.5, np.nan, 0.1, np.inf, -np.inf, 0.5], np.float32)
    lab = np.array([1, 0, 0, 1, 0, 1])
    fps, tps, thr = eng.roc_points(sc, lab)                        # nan -> 0, +-inf -> +-FLT_MAX (np.nan_to_num, utils.py:77-78)
    ofps, otps, othr = o_metrics.binary_clf_curve(lab, np.nan_to_num(sc))
    assert np.array_equal(fps, ofps) and np.array_equal(tps, otps) and np.array_equal(thr.astype(np.float64), othr)
    one = eng.roc_points(np.array([0.25], np.float32), np.array([1]))
    assert one[0].tolist() == [0.0] and one[1].tolist() == [1.0] and one[2].tolist() == [0.25]
    with pytest.raises(_lib.SvhipError, match="not 0 / 1"):
        eng.roc_points(np.array([0.1, 0.2], np.float32), np.array([0, 2]))
    with pytest.raises(ValueError):
        eng.roc_points(np.array([], np.float32), np.array([]))
    with pytest.raises(ValueError):
        eng.min_dcf(np.array([0.1, 0.2], np.float32), np.array([1]), 0.05, 1, 1)
