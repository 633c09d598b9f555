"""GPU: the headline configuration at full size (BASELINE configs[1]: ECAPA-TDNN C = 1024, bf16, 256 utterances of 2 s)
through size-independent properties — the oracle cannot run 256 utterances in seconds — and the C ABI's error behaviour."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from speakerverification_amd import _lib, synth
from speakerverification_amd.engine import Engine

pytestmark = pytest.mark.gpu


def cos_rows(a, b):
    return (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))


@pytest.fixture(scope="module")
def big():
    sd = synth.synth_state_dict(synth.ecapa_param_spec(C=1024), seed=1)
    eng = Engine(model="ecapa", compute="bf16", channels=1024, max_batch=256)
    eng.load_state_dict(sd)
    eng.finalize()
    wav = synth.synth_waveforms(256, 32000, seed=20220829)
    yield eng, sd, wav
    eng.close()


def test_full_batch_is_deterministic_and_finite(big):
    eng, _, wav = big
    a = eng.embed_wave(wav)
    b = eng.embed_wave(wav)
    assert a.shape == (256, 192) and np.isfinite(a).all()
    assert np.array_equal(a, b)                                   # same launch sequence, no atomics: bit-identical


def test_utterances_do_not_see_their_batch(big):
    """An embedding depends on its own waveform only: permuting the batch permutes the rows, and a small batch gives the
    same rows.  Not bit-exact by construction — the SE / ASP time sums are accumulated per 256-row GEMM tile, and an
    utterance's tile cut moves with its position — but far inside the bf16 path's own error (cos >= 0.999 to fp32)."""
    eng, _, wav = big
    full = eng.embed_wave(wav)
    perm = np.random.Generator(np.random.PCG64(9)).permutation(256)
    shuffled = eng.embed_wave(wav[perm])
    assert cos_rows(shuffled, full[perm]).min() >= 0.99995
    small = eng.embed_wave(wav[:8])
    assert cos_rows(small, full[:8]).min() >= 0.99995
    one = eng.embed_wave(wav[100:101])
    assert cos_rows(one, full[100:101]).min() >= 0.99995
    # the reference API's own batch (embed_utterance: num_eval = 10 - 20 crops of one file per call, src/model.py:675-704) takes the
    # small-batch routes (time-sliced Res2Net chain, 16-wave small-M linears): the same rows to bf16 round-off
    eng.profile(True)
    twenty = eng.embed_wave(wav[40:60])
    labels = eng.profile_results()
    eng.profile(False)
    assert "res2net_slices" in labels and "res2net_chain" not in labels, sorted(labels)
    assert cos_rows(twenty, full[40:60]).min() >= 0.99995


def test_bf16_full_batch_tracks_the_fp32_path(big):
    """bf16 rows of the full batch vs the fp32 (1e-4-parity) path on the same waveforms, 16 utterances."""
    eng, sd, wav = big
    full = eng.embed_wave(wav)
    f32 = Engine(model="ecapa", compute="f32", channels=1024, max_batch=16)
    f32.load_state_dict(sd)
    f32.finalize()
    ref = f32.embed_wave(wav[:16])
    f32.close()
    assert cos_rows(full[:16], ref).min() >= 0.999
    assert np.abs(full[:16] - ref).max() <= 0.03 * np.abs(ref).max()


def test_scaling_and_silence(big):
    """Log-mel + mean normalisation makes the network invariant to input gain; digital silence must stay finite."""
    eng, _, wav = big
    a = eng.embed_wave(wav[:4])
    b = eng.embed_wave(wav[:4] * 0.25)
    assert cos_rows(a, b).min() >= 0.999
    z = eng.embed_wave(np.zeros((2, 32000), np.float32))
    assert np.isfinite(z).all() and cos_rows(z[:1], z[1:]).min() >= 0.99995      # rows differ only by their tile cut


@pytest.fixture(scope="module")
def big_rn():
    sd = synth.synth_state_dict(synth.rawnet2_param_spec(), seed=1)
    eng = Engine(model="rawnet2", compute="f16", embed_dim=320, max_batch=256)
    eng.load_state_dict(sd)
    eng.finalize()
    wav = synth.synth_waveforms(256, 32000, seed=20220830)
    yield eng, sd, wav
    eng.close()


def test_rawnet2_full_batch_properties(big_rn):
    """BASELINE configs[2] at full size (B = 256: rn_block128 gives each workgroup exactly one utterance, rn_sinc runs two
    persistent workgroups per CU over 256 x N items) — until now only bench.py's isfinite() looked at this regime.
    Deterministic; rows do not depend on their place in the batch nor on the batch size; 16 rows track the fp32 engine."""
    eng, sd, wav = big_rn
    a = eng.embed_wave(wav)
    b = eng.embed_wave(wav)
    assert a.shape == (256, 320) and np.isfinite(a).all()
    assert np.array_equal(a, b)
    perm = np.random.Generator(np.random.PCG64(11)).permutation(256)
    shuffled = eng.embed_wave(wav[perm])
    assert cos_rows(shuffled, a[perm]).min() >= 0.9999
    small = eng.embed_wave(wav[:8])
    assert cos_rows(small, a[:8]).min() >= 0.9999
    f32 = Engine(model="rawnet2", compute="f32", embed_dim=320, max_batch=16)
    f32.load_state_dict(sd)
    f32.finalize()
    ref = f32.embed_wave(wav[:16])
    f32.close()
    c = cos_rows(a[:16], ref)
    print("rawnet2 f16 B=256 vs f32: cos", c.min(), "max err / scale", np.abs(a[:16] - ref).max() / np.abs(ref).max())
    assert c.min() >= 0.999                                       # the 16-bit bars of ECAPA's bf16 mode
    assert np.abs(a[:16] - ref).max() <= 0.03 * np.abs(ref).max()


def test_rawnet2_f32x3_full_batch_properties(big_rn):
    """The 1e-4 mode of RawNet2 at full size (B = 256: the split sinc kernel's persistent grid over 256 x 331 tiles, 21 166-tile launches of the
    split convolution kernel, mode 3's 84 tiles of 126 frames per utterance): deterministic, rows independent of their place in the batch
    (every reduction is per utterance in a fixed order: bit-identical under a permutation), 16 rows within 1e-4 of the exact-fp32 engine."""
    _, sd, wav = big_rn
    eng = Engine(model="rawnet2", compute="f32x3", embed_dim=320, max_batch=256)
    eng.load_state_dict(sd)
    eng.finalize()
    a = eng.embed_wave(wav)
    assert a.shape == (256, 320) and np.isfinite(a).all()
    assert np.array_equal(a, eng.embed_wave(wav))
    perm = np.random.Generator(np.random.PCG64(12)).permutation(256)
    shuffled = eng.embed_wave(wav[perm])
    assert np.array_equal(shuffled, a[perm])
    small = eng.embed_wave(wav[:8])
    eng.close()
    assert float(np.abs(small - a[:8]).max()) <= 1e-4 * float(np.abs(a).max())       # (B <= 64 takes the sliced tails: another summation order)
    f32 = Engine(model="rawnet2", compute="f32", embed_dim=320, max_batch=16)
    f32.load_state_dict(sd)
    f32.finalize()
    ref = f32.embed_wave(wav[:16])
    f32.close()
    err = float(np.abs(a[:16] - ref).max() / np.abs(ref).max())
    print("rawnet2 f32x3 B=256 vs f32: max err / scale", err)
    assert err <= 1e-4


def test_bench_shard_path_at_world_one(capsys):
    """`bench.py --config shard` (BASELINE configs[4]'s code path: utterances generated on the device, embedded in
    batches, ONE all-gather, row-sharded cosine + AS-norm scoring) was in no test.  2 000 utterances at world size 1: the record's
    own checks, and the first 64 embeddings against the counter-RNG oracle's waveforms through the fp32 (1e-4-parity) engine."""
    import argparse
    import json
    import bench
    from oracle import synthwave as o_synth
    args = argparse.Namespace(gpus=1, steps=1, warmup=1, compute="bf16", batch=256, model="ecapa", config="shard", utts_per_gpu=2000,
                              no_cpu_baseline=True, no_scoring=False, no_extras=True, allow_gloo=False, sustain_seconds=0.0)
    dev = torch.device("cuda", 0)
    ranks = bench.Ranks(args)
    with torch.cuda.stream(torch.cuda.Stream(device=dev)):
        line, shard = bench.run_shard(args, ranks, dev, keep=True)
        torch.cuda.synchronize()
    printed = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert printed["config"]["utterances_per_gpu"] == 2000 and printed["n_gpus"] == 1 and printed["scaling"] == "weak"
    assert line["finite"] and line["shard"]["own_block_intact"] and line["shard"]["scores_finite"]
    assert line["shard"]["gathered_rows"] == 2000 and line["steps"] == 8 and line["value"] > 0
    assert line["roofline"]["launches"] > 0 and 0 < line["roofline"]["frac"] < 1
    emb = shard[:64].cpu().numpy()
    wav = o_synth.synth_waveforms(bench.SEED_SHARD, 0, 64, bench.SAMPLES)
    f32 = Engine(model="ecapa", compute="f32", channels=bench.CHANNELS, max_batch=64)
    f32.load_state_dict(synth.synth_state_dict(synth.ecapa_param_spec(C=bench.CHANNELS), seed=1))
    f32.finalize()
    ref = f32.embed_wave(wav)
    f32.close()
    assert cos_rows(emb, ref).min() >= 0.999
    assert np.abs(emb - ref).max() <= 0.03 * np.abs(ref).max()


def test_bench_stdout_is_one_compact_json_line(tmp_path):
    """`python bench.py` as the driver runs it (a child process): stdout is exactly ONE line, <= 4 KB, that
    json.loads and carries the contract's fields, `roofline` and (with the CPU leg on) `cpu_baseline`; the sub-records went to the
    record file.  Short legs only (2 steps, no extras / scoring / CPU leg) — the full default run is the driver's."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rf = tmp_path / "records.jsonl"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--no-extras", "--no-scoring",
                        "--no-cpu-baseline", "--sustain-seconds", "0", "--record-file", str(rf)],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    out_lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(out_lines) == 1 and len(out_lines[0]) <= 4096
    line = json.loads(out_lines[0])
    assert line["metric"].startswith("embeddings/sec") and line["steps"] == 2 and line["warmup"] == 1 and line["n_gpus"] == 1
    assert line["dtype"] == "bf16" and "configs[1]" in line["config"]["workload"] and line["value"] > 0
    assert abs(line["value"] - 256 * 2 / (line["ms_per_step"] * 2e-3)) < 1e-3 * line["value"]
    r = line["roofline"]
    assert r["bound"] == "mfma" and r["kernel"] == "gemm_pw3" and r["launches"] == 14 and 0 < r["frac"] < 1 and r["peak"] == 2500.0
    assert line["check"]["ok"] and line["cpu_baseline"] is None
    recs = [json.loads(ln) for ln in open(rf)]
    assert [x["record"] for x in recs] == ["kernels", "headline_full"] and "gemm_pw3" in recs[0]["kernels"]


def test_c_abi_error_behaviour():
    """Status codes and messages of the boundary (include/svhip.h): never a crash, never a silent success."""
    lib = _lib.load()
    eng = Engine(model="ecapa", channels=64, max_batch=2)
    wav = synth.synth_waveforms(3, 32000)
    out = np.empty((3, 192), np.float32)
    # forward before the weights are final
    rc = lib.svhip_embed_wave(eng.h, wav.ctypes.data, 2, 32000, out.ctypes.data, 0)
    assert rc == -3 and b"finalize" in lib.svhip_last_error(eng.h)                                  # SVHIP_ERR_STATE
    # unknown tensor / wrong shape keep the reference's wording (model.py:730-742)
    bad = np.zeros((3, 3), np.float32)
    shp = (C.c_int64 * 2)(3, 3)
    assert lib.svhip_load_tensor(eng.h, b"no.such.tensor", bad.ctypes.data, shp, 2, _lib.F32) == -1
    assert b"is not in the model" in lib.svhip_last_error(eng.h)
    assert lib.svhip_load_tensor(eng.h, b"fc.conv.bias", bad.ctypes.data, shp, 2, _lib.F32) == -1
    assert b"Wrong parameter" in lib.svhip_last_error(eng.h)
    assert lib.svhip_finalize_weights(eng.h) == -6 and b"never loaded" in lib.svhip_last_error(eng.h)   # SVHIP_ERR_MISSING
    eng.load_state_dict(synth.synth_state_dict(synth.ecapa_param_spec(C=64), seed=2))
    eng.finalize()
    assert lib.svhip_finalize_weights(eng.h) == -3                                                  # twice
    # batch / length outside the handle's geometry
    with pytest.raises(_lib.SvhipError, match="max_batch"):
        eng.embed_wave(wav)
    with pytest.raises(_lib.SvhipError):
        eng.embed_wave(wav[:2, :16000])
    assert lib.svhip_embed_wave(eng.h, None, 2, 32000, out.ctypes.data, 0) == -1                    # null pointer
    assert lib.svhip_embed_wave(eng.h, wav.ctypes.data, 0, 32000, out.ctypes.data, 0) == -1         # empty batch
    # ASYNC is only defined for device pointers
    assert lib.svhip_embed_wave(eng.h, wav.ctypes.data, 2, 32000, out.ctypes.data, _lib.ASYNC) != 0
    # the handle still works after all of that
    good = eng.embed_wave(wav[:2])
    assert np.isfinite(good).all()
    # scoring argument checks
    E = np.random.default_rng(0).standard_normal((4, 192)).astype(np.float32)
    with pytest.raises(_lib.SvhipError):
        eng.score_pairs(E, np.array([0, 9], np.int32), np.array([1, 2], np.int32))                  # index out of range
    eng.close()


def test_profile_filter_brackets_only_the_named_kernel():
    """svhip_profile_filter: bench.py times the roofline kernel inside the timed region and nothing else."""
    eng = Engine(model="ecapa", compute="bf16", channels=512, max_batch=8)
    eng.load_state_dict(synth.synth_state_dict(synth.ecapa_param_spec(C=512), seed=2))
    eng.finalize()
    wav = synth.synth_waveforms(8, 32000)
    # (which kernel the 7 big GEMMs take depends on the grid: at B = 8 the persistent kernel walks their few tiles as column halves —
    #  round 5 — and what it does not take goes to the per-tile 256 x 256 kernel or gemm_pw's narrower tile)
    eng.profile(True)
    eng.embed_wave(wav)
    every = eng.profile_results()
    assert {"fbank_fused", "res2net_slices", "asp_bf16", "se_apply"} <= set(every)   # (B = 8: the time-sliced chain; bf16: the fused front-end)
    label = "gemm_pw3" if "gemm_pw3" in every else "gemm_pw2"
    assert 1 <= every[label]["launches"] <= 7
    eng.profile(True, only=label)
    eng.embed_wave(wav)
    only = eng.profile_results()
    assert set(only) == {label} and only[label]["launches"] == every[label]["launches"] and only[label]["ms"] > 0
    eng.profile(True)
    eng.embed_wave(wav)
    every = eng.profile_results()
    eng.profile(False)
    eng.embed_wave(wav)
    assert eng.profile_results() == every            # nothing recorded while profiling is off
    eng.close()


def test_f32x3_full_batch_of_short_utterances():
    """B = 256 utterances of 130 frames on an F32X3 handle: every big GEMM takes the persistent kernel WITHOUT a grid cap, but the
    utterances are shorter than its 256-row tile, so the column sums (SE squeeze, ASP statistics) must come from their own kernels
    while the operands stay in the split layout.  Finite, deterministic, and rows 0..3 equal the same utterances embedded alone."""
    C, T, B = 1024, 130, 256
    eng = Engine(model="ecapa", compute="f32x3", channels=C, max_batch=B, samples=(T - 1) * 80)
    eng.load_state_dict(synth.synth_state_dict(synth.ecapa_param_spec(C=C), seed=41))
    eng.finalize()
    mel = synth.synth_mel(B, 80, T, seed=42)
    eng.profile(True)
    a = eng.embed_features(mel)
    labels = eng.profile_results()
    b = eng.embed_features(mel)
    # (130 M-tiles: fewer than CUs, so the Res2Net steps stay on the conv kernels; the pointwise layers have 520 / 1560 tiles)
    assert "gemm_pw3x3" in labels and "gemm_pw3cv" in labels and "se_mean" in labels and "asp_gstats" in labels, labels.keys()
    assert np.isfinite(a).all() and np.array_equal(a, b)
    small = eng.embed_features(mel[:4])
    scale = float(np.abs(a).max())
    assert float(np.abs(small - a[:4]).max()) <= 1e-4 * scale
    eng.close()


@pytest.mark.parametrize("kind", ["gaussian", "speakers"])
def test_asnorm_statistics_at_the_full_trial_list_size(kind):
    """BASELINE configs[3] at its full size inside the test suite (the 1.2 M case ran only in bench.py): cohort statistics of
    1.2 M x 192 embeddings against 5 994 cohort rows, top 200, ten chunks of the fused kernel — 96 sampled rows (the first and last of
    every chunk among them) against the float64 statement; no row may need the slab path; a second call returns the same bits.  `speakers`:
    embeddings around the cohort's own centroids (same-speaker cosine 0.5 - 0.8), as trained embeddings are."""
    import torch
    from speakerverification_amd.engine import Engine
    dev = torch.device("cuda", 0)
    N, K, top, D = 1_200_000, 5994, 200, 192
    g = torch.Generator(device=dev).manual_seed(11)
    def unit(*shape):
        x = torch.randn(shape, generator=g, device=dev, dtype=torch.float32)
        return x / x.norm(dim=-1, keepdim=True)
    cohort = unit(K, D)
    if kind == "gaussian":
        E = unit(N, D)
    else:
        spk = torch.randint(0, K, (N,), generator=g, device=dev)
        a = (0.5 + 0.3 * torch.rand((N, 1), generator=g, device=dev)).sqrt()
        E = a * cohort[spk] + (1.0 - a * a).sqrt() * unit(N, D)
        E = E / E.norm(dim=1, keepdim=True)
    eng = Engine(model="none", max_batch=1)
    mu, sd = eng.asnorm_stats(E, cohort, top)
    assert eng.asnorm_last_fallback == 0 and eng.asnorm_last_refit == (0, 0)
    edges = [c * 131072 + o for c in range(10) for o in (0, 131071) if c * 131072 + o < N]
    rows = torch.tensor(sorted(set(edges + list(range(7, N, N // 70)) + [N - 1])), device=dev)
    S = (E[rows].double() @ cohort.double().T).sort(dim=1, descending=True).values[:, :top]
    emu = float((mu[rows].double() - S.mean(1)).abs().max())
    esd = float(((sd[rows].double() - S.std(1, unbiased=False)).abs() / S.std(1, unbiased=False)).max())
    print(f"{kind}: {rows.numel()} sampled rows of {N}: mu within {emu:.2e}, sigma within {esd:.2e} relative")
    assert emu <= 2e-7 and esd <= 1e-5
    mu2, sd2 = eng.asnorm_stats(E, cohort, top)
    assert torch.equal(mu2, mu) and torch.equal(sd2, sd)
    eng.close()
