"""GPU parity: ECAPA-TDNN forward through the C ABI vs the oracle / golden fixtures."""
import os

import numpy as np
import pytest
import torch

from oracle import ecapa as o_ecapa
from speakerverification_amd import synth
from speakerverification_amd.engine import Engine

pytestmark = pytest.mark.gpu

STAGES = ["blocks.0", "blocks.1", "blocks.2", "blocks.3", "mfa", "asp", "asp_bn"]


def make_engine(C, T, B, compute, seed_w):
    eng = Engine(model="ecapa", compute=compute, channels=C, max_batch=B, samples=(T - 1) * 80 if T != 401 else 32000)
    sd = synth.synth_state_dict(synth.ecapa_param_spec(C=C), seed=seed_w)
    eng.load_state_dict(sd)
    eng.finalize()
    return eng, sd


def stage_cf(eng, name, B, T):
    """library stages are frame-major (B, T, C); the reference's are (B, C, T)."""
    a = eng.get_stage(name)
    if name in ("asp", "asp_bn"):
        return a.reshape(B, -1, 1)
    return a.reshape(B, T, -1).transpose(0, 2, 1)


def test_ecapa_c64_stages_fp32(golden_dir):
    g = np.load(os.path.join(golden_dir, "ecapa_C64_T50.npz"))
    C, T, B = int(g["C"]), int(g["T"]), int(g["B"])
    eng, _ = make_engine(C, T, B, "f32", int(g["seed_w"]))
    mel = synth.synth_mel(B, 80, T, seed=int(g["seed_x"]))
    out = eng.embed_features(mel)
    for n in STAGES:
        ref = g["st_" + n]
        got = stage_cf(eng, n, B, T)
        tol = 1e-4 * max(1.0, float(np.abs(ref).max()))
        assert got.shape == ref.shape, n
        assert float(np.abs(got - ref).max()) <= tol, (n, float(np.abs(got - ref).max()), tol)
    ref = g["out"]
    assert float(np.abs(out - ref).max()) <= 1e-4 * max(1.0, float(np.abs(ref).max()))


@pytest.mark.parametrize("compute", ["f32", "f32x3"])
@pytest.mark.parametrize("C", [512, 1024])
def test_ecapa_full_fp32_matches_reference(golden_dir, C, compute):
    """f32: exact fp32 MFMA.  f32x3 (SVHIP_F32X3): fp32 storage, conv GEMM products as three bf16 MFMAs on hi / lo-split operands
    (~2^-17 per product): the SAME 1e-4 bars (measured 3.0e-5 of the scale, 6.2e-6 on normalised embeddings at C = 1024)."""
    g = np.load(os.path.join(golden_dir, f"ecapa_C{C}_T401.npz"))
    B, T = int(g["B"]), int(g["T"])
    eng, _ = make_engine(C, T, B, compute, int(g["seed_w"]))
    mel = synth.synth_mel(B, 80, T, seed=int(g["seed_x"]))
    out = eng.embed_features(mel)
    ref = g["out"]
    scale = float(np.abs(ref).max())
    err = float(np.abs(out - ref).max())
    # tolerance: 1e-4 of the embedding scale (north_star: within 1e-4 fp32; embeddings here are O(100))
    assert err <= 1e-4 * max(1.0, scale), (err, scale)
    # and ABSOLUTE 1e-4 on what scoring consumes: the L2-normalised embeddings (F.normalize, src/model.py:421-423)
    nrm = lambda a: a / np.maximum(np.linalg.norm(a, axis=1, keepdims=True), 1e-12)
    err_n = float(np.abs(nrm(out) - nrm(ref)).max())
    print(f"C={C} {compute}: max|d_emb| = {err:.3e} (scale {scale:.1f}, relative {err / scale:.2e}); L2-normalised abs err {err_n:.3e}")
    assert err_n <= 1e-4, err_n
    # stage checksums captured from the reference
    ctol = 2e-5 if compute == "f32" else 1e-4
    for n in STAGES:
        cs = g["cs_" + n]
        got = stage_cf(eng, n, B, T).astype(np.float64)
        assert abs(got.sum() - cs[0]) <= ctol * cs[1] + 1e-3, n
        assert abs(np.abs(got).sum() - cs[1]) <= ctol * cs[1] + 1e-3, n


@pytest.mark.parametrize("C", [512, 1024])
def test_ecapa_bf16_close_to_fp32_reference(golden_dir, C):
    g = np.load(os.path.join(golden_dir, f"ecapa_C{C}_T401.npz"))
    B, T = int(g["B"]), int(g["T"])
    eng, _ = make_engine(C, T, B, "bf16", int(g["seed_w"]))
    mel = synth.synth_mel(B, 80, T, seed=int(g["seed_x"]))
    out = eng.embed_features(mel)
    ref = g["out"]
    # bf16 activations/weights with fp32 accumulation: stated tolerance 3e-2 of the embedding scale,
    # cosine to the fp32 reference embedding >= 0.999 (the reference's own bf16 autocast differs from
    # its fp32 output by ~4e-3 relative)
    rel = float(np.abs(out - ref).max()) / float(np.abs(ref).max())
    cos = np.sum(out * ref, axis=1) / (np.linalg.norm(out, axis=1) * np.linalg.norm(ref, axis=1))
    print("bf16 rel err", rel, "cos", cos)
    assert rel <= 3e-2
    assert float(cos.min()) >= 0.999
    # per-stage: the fused kernels (Res2Net chain, LDS-DMA GEMMs) against the reference's stage checksums
    for n in STAGES:
        cs = g["cs_" + n]
        got = stage_cf(eng, n, B, T).astype(np.float64)
        assert abs(np.abs(got).sum() - cs[1]) <= 1e-2 * cs[1], n
        assert abs(got.sum() - cs[0]) <= 1e-2 * cs[1], n


def test_batch_invariance_and_chunking():
    """Embedding of an utterance must not depend on which batch it travels in."""
    C, T = 512, 401
    eng, _ = make_engine(C, T, 16, "f32", 1)
    mel = synth.synth_mel(16, 80, T, seed=5)
    full = eng.embed_features(mel)
    part = np.concatenate([eng.embed_features(mel[:5]), eng.embed_features(mel[5:16])])
    assert float(np.abs(full - part).max()) <= 1e-5 * float(np.abs(full).max())


def test_device_pointers_roundtrip():
    C, T = 512, 401
    eng, _ = make_engine(C, T, 4, "f32", 1)
    mel = synth.synth_mel(4, 80, T, seed=6)
    host = eng.embed_features(mel)
    dev = eng.embed_features(torch.from_numpy(mel).cuda())
    assert dev.is_cuda
    assert np.array_equal(dev.cpu().numpy(), host)


@pytest.mark.parametrize("tag,features", [("mel", "melspectrogram"), ("raw", "raw")])
def test_input_norm_and_raw_feature_variants(golden_dir, tag, features):
    """input_norm=True (InstanceNorm1d(80, affine), ECAPA_TDNN.py:406-409,477-478), with and without the
    log / mean-norm prologue (features='raw' is how the fusion models build their ECAPA branch)."""
    from speakerverification_amd.models import ECAPA_TDNN
    g = np.load(os.path.join(golden_dir, "ecapa_C64_input_norm.npz"))
    m = ECAPA_TDNN.MainModel(nOut=192, channels=[64] * 4 + [192], n_mels=80, augment=False,
                             augment_options={"augment_chain": []}, features=features, input_norm=True)
    m.load_state_dict(synth.synth_state_dict(synth.ecapa_param_spec(C=64, input_norm=True), seed=4))
    out = m(synth.synth_mel(2, 80, 50, seed=13))
    ref = g["out_" + tag]
    assert out.shape == ref.shape
    assert float(np.abs(out - ref).max()) <= 1e-4 * max(1.0, float(np.abs(ref).max()))


@pytest.mark.parametrize("L", [16000, 24000, 40000])
def test_bf16_other_utterance_lengths(L):
    """Geometries off the 2 s default take other kernel paths (attention tail with several clamped frame tiles at T = 201,
    the un-fused Res2Net / pooling fallbacks past T = 416): the bf16 path must track the fp32 path on all of them."""
    C, B = 512, 3
    sd = synth.synth_state_dict(synth.ecapa_param_spec(C=C), seed=4)
    wav = synth.synth_waveforms(B, L, seed=5)
    outs = {}
    for compute in ("f32", "bf16"):
        eng = Engine(model="ecapa", compute=compute, channels=C, max_batch=B, samples=L)
        eng.load_state_dict(sd)
        eng.finalize()
        outs[compute] = eng.embed_wave(wav)
        eng.close()
    a, b = outs["f32"], outs["bf16"]
    cos = (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))
    assert np.isfinite(b).all() and cos.min() >= 0.999, cos


@pytest.mark.parametrize("C,L,B", [(64, 8000, 3), (128, 16000, 1), (128, 33000, 2), (256, 34000, 2), (512, 4000, 3),
                                   (1024, 8000, 2), (1024, 33040, 1), (192, 32000, 2)])
def test_bf16_geometry_sweep(C, L, B):
    """Channel counts, utterance lengths and batch sizes that select every dispatch branch (GEMM kernels by N, fused /
    unfused Res2Net by C/8 and T, fused / unfused attention tail by 3C % 128 and T, column-sum epilogue by T >= 256)."""
    sd = synth.synth_state_dict(synth.ecapa_param_spec(C=C), seed=6)
    wav = synth.synth_waveforms(B, L, seed=7)
    outs = {}
    for compute in ("f32", "bf16"):
        eng = Engine(model="ecapa", compute=compute, channels=C, max_batch=B, samples=L)
        eng.load_state_dict(sd)
        eng.finalize()
        outs[compute] = eng.embed_wave(wav).reshape(B, -1)
        eng.close()
    a, b = outs["f32"], outs["bf16"]
    cos = (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))
    assert np.isfinite(b).all() and cos.min() >= 0.998, (C, L, B, cos)


@pytest.mark.parametrize("C,B,cus", [(512, 8, 5), (512, 5, 3), (1024, 6, 8), (256, 16, 4)])
def test_persistent_gemm_matches_the_per_tile_kernel(C, B, cus, monkeypatch):
    """gemm_pw3 (persistent workgroups, epilogue from the accumulators) against gemm_pw2 (one workgroup per tile, LDS output image)
    on the same layers.  SVHIP_PW3_CUS caps the persistent grid so that a small batch walks several tiles per workgroup: both
    constant strips, the relaxed and the strict vmcnt counts (first tile, tile after a masked last M-tile), utterance boundaries
    inside a wave's 128 rows and outside.  The GEMM outputs are the same arithmetic (bit-identical tdnn1 output); the column sums
    are fp32 sums of unrounded values instead of bf16-rounded ones, so everything behind an SE gate agrees to bf16 round-off."""
    T = 401
    mel = synth.synth_mel(B, 80, T, seed=31)
    outs, stages, launches = {}, {}, {}
    for mode in ("0", str(cus)):
        monkeypatch.setenv("SVHIP_PW3_CUS", mode)
        eng, _ = make_engine(C, T, B, "bf16", 7)
        eng.profile(True)
        outs[mode] = eng.embed_features(mel)
        launches[mode] = eng.profile_results()
        stages[mode] = {n: stage_cf(eng, n, B, T).astype(np.float32) for n in STAGES}
        eng.profile(False)
        eng.close()
    assert "gemm_pw3" not in launches["0"]
    assert launches[str(cus)]["gemm_pw3"]["launches"] >= 6          # tdnn1 / tdnn2 of the three blocks (+ mfa when its grid is large enough)
    a, b = outs["0"], outs[str(cus)]
    assert np.isfinite(b).all()
    cos = np.sum(a * b, axis=1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))
    assert float(cos.min()) >= 0.99999, cos
    assert float(np.abs(a - b).max()) <= 5e-3 * float(np.abs(a).max())
    for n in STAGES:
        x, y = stages["0"][n], stages[str(cus)][n]
        assert float(np.abs(x - y).max()) <= 2e-2 * max(1.0, float(np.abs(x).max())), n
        assert abs(float(np.abs(x).sum()) - float(np.abs(y).sum())) <= 1e-3 * float(np.abs(x).sum()), n


@pytest.mark.parametrize("C,B,cus", [(512, 8, 5), (1024, 6, 8), (512, 5, 3)])
def test_four_wave_gemm_matches_the_eight_wave_kernel(C, B, cus, monkeypatch):
    """Round 6: option pw4 routes the plain pointwise bf16 layers with more tiles than workgroups (tdnn1 / tdnn2 / mfa) to gemm_pw4
    (256 x 256 tiles on FOUR waves, 128 x 128 per wave, one barrier per K tile) instead of gemm_pw3.  Same products in the same k order;
    the bias joins after the last product instead of before the first, so an output may differ by one rounding of the fp32 sum before it
    is rounded to bf16.  SVHIP_PW3_CUS caps the grid so that a small batch walks several tiles per workgroup (both constant strips, the
    relaxed store counts of all three column-sum forms, a masked last M-tile, utterance boundaries inside a wave's 128 rows)."""
    T = 401
    mel = synth.synth_mel(B, 80, T, seed=31)
    monkeypatch.setenv("SVHIP_PW3_CUS", str(cus))
    eng, _ = make_engine(C, T, B, "bf16", 7)
    outs, stages, launches = {}, {}, {}
    for mode in (0, 1):
        eng.set_option("pw4", mode)
        eng.profile(True)
        outs[mode] = eng.embed_features(mel)
        launches[mode] = eng.profile_results()
        stages[mode] = {n: stage_cf(eng, n, B, T).astype(np.float32) for n in STAGES}
        eng.profile(False)
    again = eng.embed_features(mel)
    eng.close()
    assert "gemm_pw4" not in launches[0] and launches[0]["gemm_pw3"]["launches"] >= 6
    assert launches[1]["gemm_pw4"]["launches"] >= 6, launches[1].keys()
    assert np.array_equal(again, outs[1])                                   # run to run: bitwise
    a, b = outs[0], outs[1]
    assert np.isfinite(b).all()
    cos = np.sum(a * b, axis=1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))
    print(f"pw4 vs pw3, C = {C}: cos >= {cos.min():.7f}, max |d| = {np.abs(a - b).max() / np.abs(a).max():.2e} of the scale")
    assert float(cos.min()) >= 0.99999, cos
    assert float(np.abs(a - b).max()) <= 5e-3 * float(np.abs(a).max())
    for n in STAGES:
        x, y = stages[0][n], stages[1][n]
        assert float(np.abs(x - y).max()) <= 2e-2 * max(1.0, float(np.abs(x).max())), n
        assert abs(float(np.abs(x).sum()) - float(np.abs(y).sum())) <= 1e-3 * float(np.abs(x).sum()), n


@pytest.mark.parametrize("C,cus", [(512, 3), (1024, 3), (256, 2)])
def test_f32x3_persistent_gemm_matches_reference(golden_dir, C, cus, monkeypatch):
    """SVHIP_F32X3 handles run tdnn1 / tdnn2 / mfa on the persistent 256 x 256 kernel in its X3 form (operands in the S32 split
    layout: hi.hi + hi.lo + lo.hi bf16 MFMA triples, fp32 storage, exact erf GELU, column sums from the accumulators) once a
    layer has more tiles than the grid; SVHIP_PW3_CUS caps the grid so that the golden fixtures' small batch takes that route.
    Same bars as the exact fp32 path: 1e-4 of the embedding scale, 1e-4 absolute on L2-normalised embeddings, and the
    per-stage checksums of the reference."""
    monkeypatch.setenv("SVHIP_PW3_CUS", str(cus))
    if C == 256:
        sd = synth.synth_state_dict(synth.ecapa_param_spec(C=C), seed=11)
        mel = synth.synth_mel(3, 80, 401, seed=12)
        outs = {}
        for compute in ("f32", "f32x3"):
            eng = Engine(model="ecapa", compute=compute, channels=C, max_batch=3)
            eng.load_state_dict(sd)
            eng.finalize()
            eng.profile(True)
            outs[compute] = eng.embed_features(mel)
            labels = eng.profile_results()
            eng.close()
            assert ("gemm_pw3x3" in labels) == (compute == "f32x3"), labels
            assert "gemm_pw3r2" not in labels and "r2_step" not in labels      # C / 8 = 32: the Res2Net step kernels are built for 64 and 128 channels
        scale = float(np.abs(outs["f32"]).max())
        assert float(np.abs(outs["f32"] - outs["f32x3"]).max()) <= 1e-4 * scale
        return
    g = np.load(os.path.join(golden_dir, f"ecapa_C{C}_T401.npz"))
    B, T = int(g["B"]), int(g["T"])
    eng, _ = make_engine(C, T, B, "f32x3", int(g["seed_w"]))
    mel = synth.synth_mel(B, 80, T, seed=int(g["seed_x"]))
    eng.profile(True)
    out = eng.embed_features(mel)
    prof = eng.profile_results()
    eng.profile(False)
    # (one conversion pass: the zero-padded features; blocks.0 — the conv-gather form of the same kernel —, tdnn1 for its first
    #  two chunks, se_apply and the Res2Net steps write their outputs pre-split)
    r2 = "r2_step" if C == 1024 else "gemm_pw3r2"      # C / 8 = 128: the dedicated 128 x 128 kernel; 64: the R2 form of the persistent one
    assert prof["gemm_pw3x3"]["launches"] == 7 and prof[r2]["launches"] == 21 and prof["split_s32"]["launches"] == 1, prof.keys()
    assert prof["gemm_pw3cv"]["launches"] == 1 and "gemm_conv" not in prof, prof.keys()
    assert "gemm_conv_add" not in prof
    assert "se_mean" not in prof and "asp_gstats" not in prof          # the squeeze / global statistics come from the GEMM epilogue
    ref = g["out"]
    scale = float(np.abs(ref).max())
    err = float(np.abs(out - ref).max())
    nrm = lambda a: a / np.maximum(np.linalg.norm(a, axis=1, keepdims=True), 1e-12)
    err_n = float(np.abs(nrm(out) - nrm(ref)).max())
    print(f"C={C} f32x3 on the persistent kernel: max|d_emb| = {err:.3e} (scale {scale:.1f}, relative {err / scale:.2e}); L2-normalised abs err {err_n:.3e}")
    assert err <= 1e-4 * max(1.0, scale) and err_n <= 1e-4
    for n in STAGES:
        cs = g["cs_" + n]
        got = stage_cf(eng, n, B, T).astype(np.float64)
        assert abs(got.sum() - cs[0]) <= 1e-4 * cs[1] + 1e-3, n
        assert abs(np.abs(got).sum() - cs[1]) <= 1e-4 * cs[1] + 1e-3, n
    eng.close()


@pytest.mark.parametrize("T,B", [(401, 3), (130, 9), (64, 8), (33, 2)])
def test_f32x3_fused_attentive_pooling_matches_the_exact_path(T, B):
    """SVHIP_F32X3 handles pool with asp_x3_kernel (logits as split-bf16 MFMA triples, lane-local online softmax, x streamed once;
    models/ECAPA_TDNN.py:250-259): pooled mean / std against the exact-fp32 handle's GEMM + asp_pool path, element by element, on
    utterance lengths whose last 32-frame tile is full, nearly empty, or the only one, and batches that do / do not fill the
    groups of 8 utterances the workgroup -> XCD mapping deals out."""
    C = 256
    sd = synth.synth_state_dict(synth.ecapa_param_spec(C=C), seed=21)
    mel = synth.synth_mel(B, 80, T, seed=22)
    st = {}
    for compute in ("f32", "f32x3"):
        eng = Engine(model="ecapa", compute=compute, channels=C, max_batch=B, samples=(T - 1) * 80 if T != 401 else 32000)
        eng.load_state_dict(sd)
        eng.finalize()
        eng.profile(True)
        eng.embed_features(mel)
        labels = eng.profile_results()
        assert ("asp_x3" in labels) == (compute == "f32x3") and ("asp_pool" in labels) == (compute == "f32"), labels.keys()
        st[compute] = {n: eng.get_stage(n).astype(np.float64) for n in ("mfa", "asp", "asp_bn")}
        eng.close()
    for n in ("asp", "asp_bn"):
        ref, got = st["f32"][n], st["f32x3"][n]
        scale = float(np.abs(ref).max())
        err = float(np.abs(got - ref).max())
        print(f"T={T} B={B} {n}: max err {err:.3e} of scale {scale:.2f}")
        assert np.isfinite(got).all() and err <= 1e-4 * max(1.0, scale), (n, err, scale)


@pytest.mark.parametrize("C,T,B", [(512, 130, 5), (1024, 257, 3), (1024, 130, 4), (512, 401, 3)])
def test_f32x3_persistent_forms_on_ragged_geometries(monkeypatch, C, T, B):
    """every persistent-kernel form of an F32X3 handle (pointwise X3 with side outputs, Res2Net steps, conv-gather blocks.0, outputs
    kept only in the split layout) on shapes whose row count is not a multiple of the 256-row tile and whose utterances are
    shorter / longer than a tile (T < 256: no column sums from the GEMM epilogue, the squeeze and the global statistics take
    their own kernels), against the exact-fp32 handle: embeddings within 1e-4 of the scale, every stage within 1e-4 of its scale."""
    monkeypatch.setenv("SVHIP_PW3_CUS", "2")
    sd = synth.synth_state_dict(synth.ecapa_param_spec(C=C), seed=31)
    mel = synth.synth_mel(B, 80, T, seed=32)
    res = {}
    for compute in ("f32", "f32x3"):
        eng = Engine(model="ecapa", compute=compute, channels=C, max_batch=B, samples=(T - 1) * 80 if T != 401 else 32000)
        eng.load_state_dict(sd)
        eng.finalize()
        eng.profile(True)
        out = eng.embed_features(mel)
        labels = eng.profile_results()
        stages = {n: stage_cf(eng, n, B, T).astype(np.float64) for n in STAGES}
        eng.close()
        res[compute] = (out, stages, labels)
    lab = res["f32x3"][2]
    assert "gemm_pw3x3" in lab and ("r2_step" if C == 1024 else "gemm_pw3r2") in lab and "gemm_pw3cv" in lab, lab.keys()
    out32, outx3 = res["f32"][0], res["f32x3"][0]
    scale = float(np.abs(out32).max())
    assert np.isfinite(outx3).all() and float(np.abs(outx3 - out32).max()) <= 1e-4 * max(1.0, scale)
    for n in STAGES:
        a, b = res["f32"][1][n], res["f32x3"][1][n]
        assert float(np.abs(a - b).max()) <= 1e-4 * max(1.0, float(np.abs(a).max())), n


@pytest.mark.parametrize("T,B", [(401, 3), (130, 9), (64, 8), (33, 2), (500, 2)])
def test_bf16_attentive_pooling_kernels_agree(monkeypatch, T, B):
    """bf16 handles pool with asp_bf16_kernel (16 waves per CU, lane-local online softmax, moments shifted by the plain channel
    mean) unless SVHIP_ASP_V1=1 selects asp_fused_kernel (one wave per SIMD, the whole logit column in registers): same inputs,
    stages `asp` / `asp_bn` within 2e-3 of their scale (both see bf16 logits; the moments are fp32 in both), and both against the
    exact-fp32 handle within the bf16 path's bar.  T = 500 is past asp_fused's 416-frame limit: the new kernel has none."""
    C = 256
    sd = synth.synth_state_dict(synth.ecapa_param_spec(C=C), seed=51)
    mel = synth.synth_mel(B, 80, T, seed=52)
    st = {}
    for name, compute, v1 in (("f32", "f32", False), ("v2", "bf16", False), ("v1", "bf16", True)):
        if v1:
            monkeypatch.setenv("SVHIP_ASP_V1", "1")
        else:
            monkeypatch.delenv("SVHIP_ASP_V1", raising=False)
        eng = Engine(model="ecapa", compute=compute, channels=C, max_batch=B, samples=(T - 1) * 80 if T != 401 else 32000)
        eng.load_state_dict(sd)
        eng.finalize()
        eng.profile(True)
        eng.embed_features(mel)
        labels = eng.profile_results()
        if name == "v2":
            assert "asp_bf16" in labels, labels.keys()
        if name == "v1" and T <= 416:
            assert "asp_fused" in labels, labels.keys()
        st[name] = {n: eng.get_stage(n).astype(np.float64) for n in ("asp", "asp_bn")}
        eng.close()
    for n in ("asp", "asp_bn"):
        scale = float(np.abs(st["f32"][n]).max())
        e21 = float(np.abs(st["v2"][n] - st["v1"][n]).max()) / scale
        e2 = float(np.abs(st["v2"][n] - st["f32"][n]).max()) / scale
        e1 = float(np.abs(st["v1"][n] - st["f32"][n]).max()) / scale
        print(f"T={T} B={B} {n}: v2 vs v1 {e21:.2e}, v2 vs f32 {e2:.2e}, v1 vs f32 {e1:.2e}")
        assert np.isfinite(st["v2"][n]).all() and e21 <= 2e-3 and e2 <= max(3e-2, 1.5 * e1)


@pytest.mark.parametrize("T,B,cus", [(401, 3, 2), (130, 5, 1), (257, 2, 2)])
def test_f32x3_res2net_step_kernels_agree(monkeypatch, T, B, cus):
    """C / 8 = 128: the Res2Net steps of an F32X3 handle run on r2_step_kernel (128 x 128 tiles, two workgroups per CU) unless
    SVHIP_R2_BIG=1 selects the R2 form of the persistent 256 x 256 kernel: same operands, same arithmetic (hi.hi + hi.lo + lo.hi in
    the same order per 32-k block) — the two kernels add the three partial products of a 32-k block in different orders, and a last-bit difference of a chain
    value can move its `lo` half by one step of the split layout (2^-17 relative): the block outputs agree to the
    quantisation of the layout itself (measured 0.7 - 4.2e-5 of the scale, growing block by block; bar 1e-4; embeddings 3e-5),
    not to fp32 rounding.  A wrong row or tap shows up three orders above that."""
    C = 1024
    monkeypatch.setenv("SVHIP_PW3_CUS", str(cus))
    sd = synth.synth_state_dict(synth.ecapa_param_spec(C=C), seed=61)
    mel = synth.synth_mel(B, 80, T, seed=62)
    st = {}
    for name, big in (("small", False), ("big", True)):
        if big:
            monkeypatch.setenv("SVHIP_R2_BIG", "1")
        else:
            monkeypatch.delenv("SVHIP_R2_BIG", raising=False)
        eng = Engine(model="ecapa", compute="f32x3", channels=C, max_batch=B, samples=(T - 1) * 80 if T != 401 else 32000)
        eng.load_state_dict(sd)
        eng.finalize()
        eng.profile(True)
        out = eng.embed_features(mel)
        labels = eng.profile_results()
        assert ("r2_step" in labels) == (not big) and ("gemm_pw3r2" in labels) == big, labels.keys()
        st[name] = (out, {n: eng.get_stage(n).astype(np.float64) for n in ("blocks.1", "blocks.2", "blocks.3")})
        eng.close()
    errs = {}
    for n in ("blocks.1", "blocks.2", "blocks.3"):
        a, b = st["small"][1][n], st["big"][1][n]
        errs[n] = float(np.abs(a - b).max()) / max(1.0, float(np.abs(b).max()))
    errs["emb"] = float(np.abs(st["small"][0] - st["big"][0]).max()) / max(1.0, float(np.abs(st["big"][0]).max()))
    print(f"T={T} B={B} r2_step vs gemm_pw3r2:", {k: f"{v:.2e}" for k, v in errs.items()})
    assert errs["emb"] <= 3e-5 and max(errs.values()) <= 1e-4, errs


@pytest.mark.parametrize("C,T_samples,B", [(1024, 32000, 5), (512, 32000, 3), (1024, 24000, 2), (256, 32000, 4)])
def test_res2net_time_slices_match_whole_utterances(C, T_samples, B):
    """Round 4 (small batches, the reference API's own operating point: embed_utterance embeds num_eval = 10 - 20 crops of ONE file per
    call, src/model.py:675-704): the fused Res2Net chain cuts every utterance into time slices with a halo of 7 * dilation frames
    (the dependency cone of seven k = 3 stages) when a batch would leave most of the chip idle.  A core row sees the same operands in
    the same order as in the whole-utterance kernel: block outputs and embeddings must agree BIT FOR BIT (option r2_slices: 0 =
    whole utterances, 3 = forced, -1 = by batch size)."""
    sd = synth.synth_state_dict(synth.ecapa_param_spec(C=C), seed=3)
    wav = synth.synth_waveforms(B, T_samples, seed=14)
    eng = Engine(model="ecapa", compute="bf16", channels=C, max_batch=B, samples=T_samples)
    eng.load_state_dict(sd)
    eng.finalize()
    outs = {}
    for mode in (0, 3, -1):
        eng.set_option("r2_slices", mode)
        emb = eng.embed_wave(wav)
        outs[mode] = (emb.copy(), eng.get_stage("blocks.1").copy(), eng.get_stage("blocks.3").copy())
    eng.close()
    assert np.isfinite(outs[0][0]).all() and np.abs(outs[0][1]).max() > 0
    for mode in (3, -1):
        for a, b in zip(outs[0], outs[mode]):
            np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("C,L,B", [(1024, 32000, 5), (512, 32000, 20), (1024, 2560, 9), (256, 10320, 7), (64, 8000, 6), (128, 12000, 3)])
def test_n128_attention_gemm_matches_the_generic_route(C, L, B):
    """Round 4: asp.tdnn (N = 128, K = 3C, per-utterance bias, ReLU -> BN -> tanh; ECAPA_TDNN.py:245-250) runs on gemm_n128 (128 x 128
    tiles, four waves, two workgroups per CU) at every batch size; option n128_off keeps gemm_pw's 256 x 128 tile.  Same products in
    the same K order; the epilogues differ in how tanh is evaluated (1 - 2 / (1 + e^2x) on the fast units against tanhf), i.e. by at
    most a bf16 rounding of a few att elements: pooled statistics and embeddings agree far inside the bf16 path's own error.
    Geometries: several tiles per utterance, short utterances (T = 33: four utterances inside one 128-row tile), a ragged last tile;
    C = 64 / 128 (round 6): K = 192 / 384 — three and six K tiles, the short ends of the kernel's three-slot X ring."""
    sd = synth.synth_state_dict(synth.ecapa_param_spec(C=C), seed=5)
    wav = synth.synth_waveforms(B, L, seed=15)
    eng = Engine(model="ecapa", compute="bf16", channels=C, max_batch=B, samples=L)
    eng.load_state_dict(sd)
    eng.finalize()
    res = {}
    for off in (1, 0):
        eng.set_option("n128_off", off)
        eng.profile(True)
        emb = eng.embed_wave(wav)
        labels = set(eng.profile_results())
        eng.profile(False)
        res[off] = (emb.copy(), eng.get_stage("asp").copy(), labels)
    eng.close()
    assert "gemm_n128" in res[0][2] and "gemm_n128" not in res[1][2], (sorted(res[0][2]), sorted(res[1][2]))
    a, b = res[0][0], res[1][0]
    cos = (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))
    rel = float(np.abs(a - b).max() / np.abs(b).max())
    srel = float(np.abs(res[0][1] - res[1][1]).max() / np.abs(res[1][1]).max())
    print(f"C={C} L={L} B={B}: gemm_n128 vs gemm_pw: embedding cos {cos.min():.7f}, max diff / scale {rel:.2e}; pooled stats {srel:.2e}")
    assert np.isfinite(a).all() and cos.min() >= 0.99999 and rel <= 5e-3 and srel <= 5e-3


def test_f32x3_input_range_guard():
    """since round 4 the split (f32x3) operands are IEEE-half hi | lo planes — fp16 dynamic range.  The network
    input is the one operand the checkpoint does not bound: non-log mel power (`features: raw`, no input_norm) of un-normalised audio
    goes straight into blocks.0.  The prologue now checks it on every call: |x| > 65504 -> SVHIP_ERR_RANGE (the call completes, the
    values were clamped); the same features on the exact-fp32 handle are simply embedded.  In-range features: no status, and the two
    handles agree to the mode's 1e-4."""
    from speakerverification_amd import _lib
    C, T, B = 64, 50, 2
    sd = synth.synth_state_dict(synth.ecapa_param_spec(C=C), seed=3)
    mel = np.abs(synth.synth_mel(B, 80, T, seed=13)).astype(np.float32) + np.float32(0.01)      # "mel power"

    def engine(compute, **kw):
        e = Engine(model="ecapa", compute=compute, channels=C, max_batch=B, samples=(T - 1) * 80, log_input=False, **kw)
        e.load_state_dict(sd)
        e.finalize()
        return e

    x3, f32 = engine("f32x3"), engine("f32")
    a, b = x3.embed_features(mel), f32.embed_features(mel)
    assert float(np.abs(a - b).max()) <= 1e-4 * float(np.abs(b).max()) and x3.numeric_status() == 0
    big = mel * np.float32(1.0e6)                                     # what int16-scaled samples would produce
    with pytest.raises(_lib.SvhipNumericError) as ei:
        x3.embed_features(big)
    assert ei.value.code == _lib.ERR_RANGE and "65504" in str(ei.value)
    assert np.isfinite(f32.embed_features(big)).all() and f32.numeric_status() == 0
    # with the log prologue (the reference's default features) the same input is in range again: log(1e6 x) - mean_t
    xl = Engine(model="ecapa", compute="f32x3", channels=C, max_batch=B, samples=(T - 1) * 80, log_input=True)
    xl.load_state_dict(sd)
    xl.finalize()
    assert np.isfinite(xl.embed_features(big)).all() and xl.numeric_status() == 0
    for e in (x3, f32, xl):
        e.close()


@pytest.mark.parametrize("mag", [1.0e7, 3.0e-9, 1.0])
def test_f32x3_first_convolution_takes_input_of_any_magnitude(mag):
    """Round 6: at the real channel counts (C % 256 == 0) blocks.0 of an F32X3 handle runs on the persistent conv-gather
    kernel, whose A operand is now s * x with s an exact power of two chosen ON THE DEVICE from the input's max |x| (s = 1 in the ordinary
    range: the arithmetic of rounds 4 - 5 bit for bit); the accumulators start at s * bias and are multiplied back before the activation.
    Non-log features a million times beyond the half-precision planes' 65504 (mel power of int16-scaled audio), or far below their
    resolution, embed as on the exact-fp32 handle instead of ending in SVHIP_ERR_RANGE — given a first layer whose weights fit such
    features (here: scaled by 1 / mag, as training on them would have)."""
    C, T, B = 512, 50, 3
    sd = synth.synth_state_dict(synth.ecapa_param_spec(C=C), seed=3)
    sd["blocks.0.conv.conv.weight"] = (sd["blocks.0.conv.conv.weight"].astype(np.float64) / mag).astype(np.float32)
    mel = ((np.abs(synth.synth_mel(B, 80, T, seed=13)) + 0.01) * mag).astype(np.float32)
    out = {}
    for compute in ("f32x3", "f32"):
        e = Engine(model="ecapa", compute=compute, channels=C, max_batch=B, samples=(T - 1) * 80, log_input=False)
        e.load_state_dict(sd)
        e.finalize()
        e.profile(True)
        out[compute] = e.embed_features(mel).copy()
        labels = set(e.profile_results())
        assert e.numeric_status() == 0
        if compute == "f32x3":
            assert {"gemm_pw3cv", "in_scale", "split_s32"} <= labels, labels
            again = e.embed_features(mel)
            assert np.array_equal(again, out[compute])
        e.close()
    a, b = out["f32x3"], out["f32"]
    err = float(np.abs(a - b).max() / np.abs(b).max())
    print(f"input magnitude {mag:g}: f32x3 against exact f32: {err:.2e} of the embedding scale")
    assert np.isfinite(a).all() and err <= 1e-4


# (cases in which every big GEMM is on the persistent kernel in BOTH runs: a 16-bit layer whose tile count lies in (CUs / 2, CUs] takes the
#  per-tile kernel by default, whose column sums are cut into other row groups)
@pytest.mark.parametrize("compute,C,L,B", [("bf16", 1024, 32000, 5), ("bf16", 512, 32000, 10), ("f32x3", 1024, 32000, 5),
                                            ("f32x3", 512, 16000, 12), ("bf16", 1024, 32000, 70)])
def test_column_halves_of_the_persistent_gemm_are_bit_identical_to_whole_tiles(compute, C, L, B):
    """Round 5: the persistent GEMM walks the last partial round of its grid — and every tile of a grid that fills at most half the chip
    (small batches) — as column HALVES of its 256 x 256 tiles (gemm_pw3.hip, "Tail split"): phases 0 and 3 of the four-phase K tile, the
    epilogue for j < 2, the second half on a channel origin 32 higher.  Every output element and every column sum sees the same products
    in the same order as in a whole tile, so the embeddings must be the same BITS with the option off (round 4's schedule), on 16-bit and
    on split (f32x3, with its S32 side outputs) handles; B = 70 at C = 1024 leaves 71 x 4 = 284 tiles = one whole round + 28 tail tiles."""
    eng = Engine(model="ecapa", compute=compute, channels=C, max_batch=B, samples=L)
    eng.load_state_dict(synth.synth_state_dict(synth.ecapa_param_spec(C=C), seed=5))
    eng.finalize()
    wav = synth.synth_waveforms(B, L, seed=23)
    eng.profile(True)
    a = eng.embed_wave(wav).copy()
    labels = set(eng.profile_results())
    eng.profile(False)
    # the whole-tile walk of the SAME kernel: halves off and the grid capped at three workgroups (without the cap a small grid would
    # leave the persistent kernel for the per-tile ones, whose column sums are cut into other row groups: equal to bf16 rounding, not bits)
    eng.set_option("pw3_tail_off", 1)
    eng.set_option("pw3_cus", 3)
    b = eng.embed_wave(wav).copy()
    eng.close()
    assert ("gemm_pw3" in labels) or ("gemm_pw3x3" in labels), labels          # the persistent kernel did run
    assert np.isfinite(a).all() and np.array_equal(a, b)


@pytest.mark.parametrize("compute", ["bf16", "f32", "f32x3"])
def test_ecapa_reports_a_nonfinite_input(compute):
    """The status word of svhip.h v4 on ECAPA handles: a NaN in one utterance's features makes that utterance's embedding NaN (the SE
    squeeze and the statistics pooling spread it over the whole utterance, as they do in the reference); the call says so
    (SVHIP_ERR_NONFINITE; on an f32x3 handle the range guard of the prologue names the input first: SVHIP_ERR_RANGE), the other rows are the
    bits they are without it, and the next clean call is clean."""
    from speakerverification_amd import _lib
    C, T, B = 64, 50, 3
    eng = Engine(model="ecapa", compute=compute, channels=C, max_batch=B, samples=(T - 1) * 80, on_numeric="ignore")
    eng.load_state_dict(synth.synth_state_dict(synth.ecapa_param_spec(C=C), seed=3))
    eng.finalize()
    mel = np.abs(synth.synth_mel(B, 80, T, seed=13)).astype(np.float32) + np.float32(0.01)
    clean = eng.embed_features(mel).copy()
    assert eng.numeric_status() == 0
    bad = mel.copy()
    bad[1, 7, 20] = np.nan
    got = np.empty_like(clean)
    rc = eng.lib.svhip_embed_features(eng.h, bad.ctypes.data, B, T, got.ctypes.data, 0)
    assert rc == (_lib.ERR_RANGE if compute == "f32x3" else _lib.ERR_NONFINITE), (rc, eng.lib.svhip_last_error(eng.h))
    assert not np.isfinite(got[1]).any()
    if compute != "bf16":                     # (a bf16 row moves by round-off with its neighbours' column sums; the fp32-grade rows do not)
        assert np.array_equal(got[[0, 2]], clean[[0, 2]])
    else:
        assert np.isfinite(got[[0, 2]]).all()
    assert eng.lib.svhip_embed_features(eng.h, mel.ctypes.data, B, T, got.ctypes.data, 0) == 0 and np.array_equal(got, clean)
    eng.close()
