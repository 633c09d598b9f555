"""CPU: the oracle restatement vs fixtures captured from the imported reference (oracle/make_golden.py)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import ecapa as o_ecapa
from oracle import fbank as o_fbank
from oracle import rawnet2 as o_rawnet2
from oracle import scoring as o_scoring
from speakerverification_amd import synth

STAGES = ["blocks.0", "blocks.1", "blocks.2", "blocks.3", "mfa", "asp", "asp_bn"]


def checksum(t):
    t = t.detach().double()
    return np.array([float(t.sum()), float(t.abs().sum())] + [float(v) for v in t.flatten()[:8]])


def test_param_specs_match_reference_state_dicts(golden_dir):
    ref = json.load(open(os.path.join(golden_dir, "param_specs.json")))
    for key, C in (("ecapa_C64", 64), ("ecapa_C512", 512), ("ecapa_C1024", 1024)):
        mine = [[n, list(s)] for n, s in synth.ecapa_param_spec(C=C)]
        assert mine == ref[key]
        assert len(mine) == 231
    mine = [[n, list(s)] for n, s in synth.rawnet2_param_spec(nOut=320)]
    assert mine == ref["rawnet2"]
    assert len(mine) == 147


def test_synth_weights_are_deterministic():
    a = synth.synth_state_dict(synth.ecapa_param_spec(C=64), seed=3)
    b = synth.synth_state_dict(synth.ecapa_param_spec(C=64), seed=3)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    assert abs(float(a["blocks.0.conv.conv.weight"].std()) - (2.0 / 400) ** 0.5) < 0.01


def test_ecapa_c64_full_stages(golden_dir):
    g = np.load(os.path.join(golden_dir, "ecapa_C64_T50.npz"))
    sd = o_ecapa.to_torch_sd(synth.synth_state_dict(synth.ecapa_param_spec(C=64), seed=int(g["seed_w"])))
    mel = torch.from_numpy(synth.synth_mel(int(g["B"]), 80, int(g["T"]), seed=int(g["seed_x"])))
    st = {}
    with torch.no_grad():
        out = o_ecapa.ecapa_forward(mel, sd, stages=st)
    for n in STAGES:
        assert float((st[n] - torch.from_numpy(g["st_" + n])).abs().max()) <= 1e-5 * max(1.0, float(np.abs(g["st_" + n]).max())), n
    assert float((out - torch.from_numpy(g["out"])).abs().max()) <= 1e-5 * float(np.abs(g["out"]).max())


@pytest.mark.parametrize("C", [512, 1024])
def test_ecapa_full_size(golden_dir, C):
    g = np.load(os.path.join(golden_dir, f"ecapa_C{C}_T401.npz"))
    sd = o_ecapa.to_torch_sd(synth.synth_state_dict(synth.ecapa_param_spec(C=C), seed=int(g["seed_w"])))
    mel = torch.from_numpy(synth.synth_mel(int(g["B"]), 80, int(g["T"]), seed=int(g["seed_x"])))
    st = {}
    with torch.no_grad():
        out = o_ecapa.ecapa_forward(mel, sd, stages=st)
    assert float((out - torch.from_numpy(g["out"])).abs().max()) <= 1e-5 * float(np.abs(g["out"]).max())
    for n in STAGES:
        cs, ref = checksum(st[n]), g["cs_" + n]
        assert abs(cs[0] - ref[0]) <= 1e-6 * ref[1] + 1e-4, n
        assert np.allclose(cs[2:], ref[2:], rtol=1e-4, atol=1e-5), n


def test_ecapa_batch1_squeeze():
    """ECAPA_TDNN.py:500: squeeze() turns a batch of one into a (nOut,) vector."""
    sd = o_ecapa.to_torch_sd(synth.synth_state_dict(synth.ecapa_param_spec(C=64), seed=3))
    with torch.no_grad():
        out = o_ecapa.ecapa_forward(torch.from_numpy(synth.synth_mel(1, 80, 50)), sd)
    assert out.shape == (192,)


def test_rawnet2(golden_dir):
    g = np.load(os.path.join(golden_dir, "rawnet2.npz"))
    sd = o_ecapa.to_torch_sd(synth.synth_state_dict(synth.rawnet2_param_spec(nOut=320), seed=int(g["seed_w"])))
    x = torch.from_numpy(synth.synth_waveforms(int(g["B"]), 32000, seed=int(g["seed_x"])))
    st = {}
    with torch.no_grad():
        out = o_rawnet2.rawnet2_forward(x, sd, stages=st)
    assert out.shape == (2, 320)
    assert float((out - torch.from_numpy(g["out"])).abs().max()) <= 1e-5 * float(np.abs(g["out"]).max())
    assert float((st["sinc_filters"][[0, 1, 63, 127]] - torch.from_numpy(g["sinc_filters_rows"])).abs().max()) <= 1e-6
    shapes = {"front": 10583, "layer1": 3527, "layer2": 1175, "layer3": 391, "layer4": 130, "layer5": 43, "layer6": 14}
    for n, t in shapes.items():
        assert st[n].shape[-1] == t, n          # SURVEY §2.3 R2-R3 time axis
    for n in ("layer1", "layer6"):
        cs, ref = checksum(st[n]), g["cs_" + n]
        assert abs(cs[0] - ref[0]) <= 1e-5 * ref[1] + 1e-3


def test_preemphasis(golden_dir):
    g = np.load(os.path.join(golden_dir, "preemphasis.npz"))
    assert np.array_equal(o_fbank.pre_emphasis(torch.from_numpy(g["a"])).numpy(), g["ya"])
    assert np.allclose(g["ya"], [[-0.97, 1.0, 1.03, 1.06, 1.09, 1.12]])
    assert np.array_equal(o_fbank.pre_emphasis(torch.from_numpy(g["b"])).numpy(), g["yb"])


def test_scoring(golden_dir):
    g = np.load(os.path.join(golden_dir, "scoring.npz"))
    top = int(g["top"])
    for i in range(g["R"].shape[0]):
        r = torch.nn.functional.normalize(torch.from_numpy(g["R"][i]), p=2, dim=1)
        c = torch.nn.functional.normalize(torch.from_numpy(g["C"][i]), p=2, dim=1)
        assert abs(o_scoring.cosine_similarity(r, c) - g["cosine"][i]) < 1e-6
        assert abs(o_scoring.cosine_similarity(torch.from_numpy(g["R"][i]), torch.from_numpy(g["C"][i])) - g["cosine_raw"][i]) < 1e-6
        assert abs(o_scoring.zt_norm_similarity(r.numpy(), c.numpy(), g["cohort"], top) - g["zt_norm"][i]) < 1e-5
        assert abs(o_scoring.zt_norm_similarity(r.numpy(), c.numpy(), g["cohort"]) - g["zt_norm_default_top"][i]) < 1e-5
        assert abs(o_scoring.pnorm_similarity(r, c) - g["pnorm"][i]) < 1e-6
    # pnorm_similarity's `p` argument (utils.py:167) at values the reference's own call site never passes: tests/golden/pnorm_p.npz
    gp = np.load(os.path.join(golden_dir, "pnorm_p.npz"))
    for k, pv in enumerate(gp["p"]):
        for i in range(gp["R"].shape[0]):
            want = gp["pnorm"][k, i]
            got = o_scoring.pnorm_similarity(torch.from_numpy(gp["R"][i]), torch.from_numpy(gp["C"][i]), p=float(pv))
            assert abs(got - want) <= 1e-6 * max(1.0, abs(want)), (pv, i)
    # batched GEMM statement on crop means == the reference's per-trial loop
    Rn = torch.nn.functional.normalize(torch.from_numpy(g["R"]), p=2, dim=2).numpy()
    Cn = torch.nn.functional.normalize(torch.from_numpy(g["C"]), p=2, dim=2).numpy()
    n = Rn.shape[0]
    E = np.concatenate([Rn.mean(axis=1), Cn.mean(axis=1)])
    got = o_scoring.asnorm_pairs(E, np.arange(n), np.arange(n, 2 * n), g["cohort"], top)
    assert np.abs(got - g["zt_norm"]).max() < 1e-5


def test_crop(golden_dir):
    g = np.load(os.path.join(golden_dir, "crop.npz"))
    rng = np.random.Generator(np.random.PCG64(77))
    for name in ("long", "short", "exact", "long10"):
        n, ne = int(g[name + "_len"]), int(g[name + "_num_eval"])
        a = (0.3 * rng.standard_normal(n)).astype(np.float32)
        got = o_scoring.crop_eval(a, 32000, ne)
        assert got.shape == (ne, 32000)
        assert np.array_equal(got[:, :4], g[name + "_first"]) and np.array_equal(got[:, -4:], g[name + "_last"])
        assert abs(float(got.astype(np.float64).sum()) - g[name + "_cs"][0]) < 1e-6 * g[name + "_cs"][1] + 1e-9


def test_fbank_restatement_is_self_consistent():
    """F2-F3 are parity-unpinned (nnAudio absent): check the restatement against an independent
    float64 rFFT statement of the same definition, and its fp32 arithmetic against fp64."""
    wav = synth.synth_waveforms(2)
    r32 = o_fbank.melspectrogram(torch.from_numpy(wav)).numpy()
    r64 = o_fbank.melspectrogram(torch.from_numpy(wav).double()).numpy()
    assert r32.shape == (2, 80, 401)
    y = o_fbank.pre_emphasis(torch.from_numpy(wav).double()).numpy()
    w, lpad = o_fbank.window_taps()
    assert lpad == 156 and np.count_nonzero(w) == 200
    mb = o_fbank.mel_basis().astype(np.float64)
    assert mb.shape == (80, 257) and int((mb > 0).sum(axis=1).max()) <= 17
    for b in range(2):
        yp = np.pad(y[b], (256, 256), mode="reflect")
        frames = np.stack([yp[i * 80:i * 80 + 512] * w.astype(np.float64) for i in range(401)])
        mel = mb @ (np.abs(np.fft.rfft(frames, axis=1)) ** 2).T
        assert np.abs(mel - r64[b]).max() <= 1e-6 * np.abs(r64[b]).max()
    peak = np.abs(r64).max(axis=(1, 2), keepdims=True)
    assert (np.abs(r32 - r64) / peak).max() <= 2e-6
    l32 = o_fbank.log_mean_norm(torch.from_numpy(r32).double()).numpy()
    l64 = o_fbank.log_mean_norm(torch.from_numpy(r64)).numpy()
    assert np.abs(l32 - l64).max() <= 1e-4


@pytest.mark.parametrize("case", ["small", "ties", "distinct", "skewed"])
def test_metrics_oracle_matches_reference_outputs(golden_dir, case):
    """oracle/metrics.py == the reference's tuneThresholdfromScore / ComputeErrorRates / ComputeMinDcf (utils.py:74-121,
    221-275, run with the installed sklearn by oracle/make_golden.py::golden_metrics), bit for bit."""
    from oracle import metrics as o_metrics
    from tests.metrics_data import metrics_case
    g = np.load(os.path.join(golden_dir, "metrics.npz"))
    sc, lab = metrics_case(case)
    res = o_metrics.tune_threshold_from_score([float(v) for v in sc], [int(v) for v in lab], [1, 0.1], [5])
    assert np.array_equal(np.array([res["gmean"][0], res["gmean"][1], res["gmean"][2]], np.float64), g[case + "_gmean"])
    assert np.array_equal(np.array(res["roc"][0], np.float64), g[case + "_tuned"])
    assert np.array_equal(np.array([res["roc"][1], res["roc"][2], res["roc"][3]], np.float64), g[case + "_eer_auc_thr"])
    assert np.array_equal(res["prec_recall"][0], g[case + "_precision"]) and np.array_equal(res["prec_recall"][1], g[case + "_recall"])
    fnrs, fprs, thr = o_metrics.compute_error_rates(sc, lab)
    assert np.array_equal(fnrs, g[case + "_fnrs"]) and np.array_equal(fprs, g[case + "_fprs"]) and np.array_equal(thr, g[case + "_thr"])
    d = g[case + "_mindcf"]
    assert o_metrics.compute_min_dcf(fnrs, fprs, thr, 0.05, 1, 1) == (d[0], d[1])
    assert o_metrics.compute_min_dcf(fnrs, fprs, thr, 0.01, 10, 1) == (d[2], d[3])


def test_philox_known_answers_and_synthetic_stream():
    """oracle/synthwave.py: Philox4x32-10 against the Random123 known-answer vectors; the waveform stream is a pure function
    of (seed, utterance, sample) with the 0.1 * N(0, 1) statistics the embedding benches assume."""
    from oracle.synthwave import philox4x32_10, synth_waveforms
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        assert philox4x32_10(np.array(ctr, np.uint32), key).tolist() == list(want)
    w = synth_waveforms(5, 3, 2, 32000)
    assert w.dtype == np.float32 and w.shape == (2, 32000) and np.abs(w).max() <= 1.0
    assert abs(float(w.mean())) < 2e-3 and abs(float(w.std()) - 0.1) < 2e-3
    assert np.array_equal(synth_waveforms(5, 4, 1, 32000)[0], w[1])          # block boundaries do not matter
    assert not np.array_equal(synth_waveforms(6, 3, 1, 32000)[0], w[0])


def test_fbank_definition_of_record_is_frozen(golden_dir):
    """VERDICT r1 #6c: F2-F3 stay parity-UNPINNED (no reference-held vector exists), but the definition of record is frozen:
    oracle/fbank.py must reproduce the committed arrays (oracle/make_fbank_fixture.py), so an edit of the restatement cannot
    move the target the HIP kernel is checked against."""
    from oracle.make_fbank_fixture import impulse_wave
    g = np.load(os.path.join(golden_dir, "fbank.npz"))
    for name, wav in (("white", synth.synth_waveforms(1)), ("speech", synth.synth_speechlike(1)), ("impulse", impulse_wave())):
        r64 = o_fbank.melspectrogram(torch.from_numpy(wav).double()).numpy()
        r32 = o_fbank.melspectrogram(torch.from_numpy(wav)).numpy()
        peak = float(np.abs(g[name + "_f64"]).max())
        assert r64.shape == g[name + "_f64"].shape
        assert np.abs(r64 - g[name + "_f64"]).max() <= 1e-12 * peak, name
        assert np.abs(r32 - g[name + "_f32"]).max() <= 2e-6 * peak, name      # fp32 conv summation order may differ across hosts


def test_oracle_reproduces_the_speaker_fixture_end_to_end(golden_dir, tmp_path):
    """tests/golden/e2e_speakers.npz (the reference's ModelHandling on four synthetic speakers, asp_bn calibrated to the files:
    cosine scores 0.05 - 0.78) against the oracle chain on the CPU: WAV -> eval-mode crops (speakerverification_amd.audio, pinned by
    crop.npz) -> oracle fbank -> oracle ECAPA -> L2-normalise -> mean |cos| over aligned crops.  Pins the oracle (not the product)
    on a fixture whose scores are sensitive to file order, cropping and crop means."""
    from speakerverification_amd import audio
    from tests.e2e_data import E2E2_SEED_W, make_e2e_speaker_files
    g = np.load(os.path.join(golden_dir, "e2e_speakers.npz"))
    files, trial_path, lines = make_e2e_speaker_files(str(tmp_path))
    sd = synth.synth_state_dict(synth.ecapa_param_spec(C=512), seed=E2E2_SEED_W)
    sd["asp_bn.norm.running_mean"] = g["asp_bn_running_mean"]
    sd["asp_bn.norm.running_var"] = g["asp_bn_running_var"]
    tsd = o_ecapa.to_torch_sd(sd)
    torch.set_num_threads(8)
    embs = {}
    with torch.no_grad():
        for f in files:
            crops = audio.loadWAV(f, {"sample_rate": 16000, "sentence_len": 2.0}, evalmode=True, num_eval=2)
            mel = o_fbank.melspectrogram(torch.from_numpy(np.asarray(crops, np.float32)))
            embs[f] = o_ecapa.ecapa_forward(mel, tsd)
    got = np.stack([embs[f].numpy() for f in files])
    assert float(np.abs(got - g["embeddings_ne2"]).max()) <= 1e-4 * float(np.abs(g["embeddings_ne2"]).max())
    sc = []
    for ln in lines:
        _, a, b = ln.split()
        r = torch.nn.functional.normalize(embs[a], p=2, dim=1)
        c = torch.nn.functional.normalize(embs[b], p=2, dim=1)
        sc.append(o_scoring.cosine_similarity(r, c))
    assert float(np.abs(np.array(sc) - g["scores_ne2"]).max()) <= 1e-5
    assert g["scores_ne2"].max() - g["scores_ne2"].min() >= 0.3
