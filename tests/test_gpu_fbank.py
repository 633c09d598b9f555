"""GPU parity: pre-emphasis + mel front-end through the C ABI vs the oracle (nnAudio formulation)."""
import numpy as np
import pytest
import torch

from oracle import fbank as o_fbank
from speakerverification_amd import synth
from speakerverification_amd.engine import Engine

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    return Engine(model="none", max_batch=8)


@pytest.mark.parametrize("kind", ["white", "speechlike"])
def test_fbank_matches_oracle(eng, kind):
    wav = synth.synth_waveforms(4) if kind == "white" else synth.synth_speechlike(4)
    got = eng.fbank(wav)
    ref32 = o_fbank.melspectrogram(torch.from_numpy(wav)).numpy()
    ref64 = o_fbank.melspectrogram(torch.from_numpy(wav).double()).numpy()
    assert got.shape == ref32.shape == (4, 80, 401)
    # power mel: relative to each utterance's peak mel power
    peak = np.abs(ref64).max(axis=(1, 2), keepdims=True)
    e_gpu = float((np.abs(got - ref64) / peak).max())
    e_cpu = float((np.abs(ref32 - ref64) / peak).max())
    print(kind, "max rel-to-peak error: gpu", e_gpu, "cpu-fp32 oracle", e_cpu)
    assert e_gpu <= 2e-6
    # what the model consumes: log(mel + 1e-6) - mean_t  (ECAPA_TDNN.py:473-476); tolerance 1e-4
    lg = o_fbank.log_mean_norm(torch.from_numpy(got).double()).numpy()
    lr = o_fbank.log_mean_norm(torch.from_numpy(ref64)).numpy()
    e_log = float(np.abs(lg - lr).max())
    e_log_cpu = float(np.abs(o_fbank.log_mean_norm(torch.from_numpy(ref32).double()).numpy() - lr).max())
    print(kind, "log-mel max abs error: gpu", e_log, "cpu-fp32 oracle", e_log_cpu)
    assert e_log <= 1e-4


def test_fbank_edges_and_device(eng):
    # impulse at the borders exercises the reflect padding of both PreEmphasis and the STFT
    wav = np.zeros((3, 32000), np.float32)
    wav[0, 0] = 1.0
    wav[1, -1] = 1.0
    wav[2, 1] = -0.5
    got = eng.fbank(torch.from_numpy(wav).cuda()).cpu().numpy()
    ref = o_fbank.melspectrogram(torch.from_numpy(wav).double()).numpy()
    assert float(np.abs(got - ref).max()) <= 2e-6 * float(np.abs(ref).max())


def test_fbank_matches_frozen_fixture(eng, golden_dir):
    """the HIP front-end against the COMMITTED definition of record (tests/golden/fbank.npz), not the live oracle"""
    import os
    from oracle.make_fbank_fixture import impulse_wave
    g = np.load(os.path.join(golden_dir, "fbank.npz"))
    for name, wav in (("white", synth.synth_waveforms(1)), ("speech", synth.synth_speechlike(1)), ("impulse", impulse_wave())):
        got = eng.fbank(wav)
        ref = g[name + "_f64"]
        peak = np.abs(ref).max(axis=(1, 2), keepdims=True)
        assert float((np.abs(got - ref) / peak).max()) <= 2e-6, name
        if name != "impulse":        # (an impulse leaves most frames at exactly zero power: log(1e-6) differences are meaningless there)
            lg = o_fbank.log_mean_norm(torch.from_numpy(got).double()).numpy()
            assert float(np.abs(lg - g[name + "_logmel_f64"]).max()) <= 1e-4, name


@pytest.mark.parametrize("kind", ["white", "speechlike"])
def test_fbank_bf16x3_split_path(kind):
    """bf16-compute handles run the DFT as hi/lo-split bf16 MFMAs (3 products): stated tolerance
    5e-3 on log-mel (measured 1.6e-3 on white noise, 1.2e-4 on speech-like input; mel power within 4e-6
    of the utterance peak), i.e. below the bf16 rounding (2^-8 relative) of the features it feeds."""
    eng = Engine(model="none", compute="bf16", max_batch=4)
    wav = synth.synth_waveforms(4) if kind == "white" else synth.synth_speechlike(4)
    got = eng.fbank(wav)
    ref64 = o_fbank.melspectrogram(torch.from_numpy(wav).double()).numpy()
    lg = o_fbank.log_mean_norm(torch.from_numpy(got).double()).numpy()
    lr = o_fbank.log_mean_norm(torch.from_numpy(ref64)).numpy()
    e_log = float(np.abs(lg - lr).max())
    peak = np.abs(ref64).max(axis=(1, 2), keepdims=True)
    print(kind, "bf16x3 log-mel max abs error", e_log, "rel-to-peak", float((np.abs(got - ref64) / peak).max()))
    assert e_log <= 5e-3


@pytest.mark.parametrize("compute", ["f32", "bf16"])
@pytest.mark.parametrize("L", [32000, 512, 5200, 10320, 15439])
def test_fbank_64_frame_kernel_equals_the_32_frame_kernel(monkeypatch, compute, L):
    """the 64-frame workgroup (one bin pair per wave, basis streamed once per 64 frames) against the 32-frame kernel it replaced
    (option fbank32), on lengths whose last tile holds one frame tile, two, or a single frame: the exact-fp32 form multiplies the
    same taps in the same order and must agree BIT FOR BIT; the split form differs only in |X|^2 = re^2 + im^2 vs sqrt-then-square."""
    eng = Engine(model="none", compute=compute, max_batch=5, samples=L)
    rng = np.random.default_rng(L)
    wav = (rng.standard_normal((5, L)) * 0.1).astype(np.float32)
    new = eng.fbank(wav)
    eng.set_option("fbank32", 1)
    old = eng.fbank(wav)
    eng.set_option("fbank32", 0)
    assert new.shape == old.shape and np.isfinite(new).all()
    if compute == "f32":
        np.testing.assert_array_equal(new, old)
    else:
        np.testing.assert_allclose(new, old, rtol=2e-6, atol=1e-12)
    eng.close()


@pytest.mark.parametrize("kind", ["white", "speechlike"])
def test_fbank_six_product_form_on_f32x3_handles(kind):
    """F32X3 handles run the DFT on six bf16 MFMAs per product block (samples and basis split EXACTLY into three bf16 parts, the
    three partial products below 2^-26 dropped): fp32-grade — held to the SAME bars as the exact-fp32 MFMA form (mel power within
    2e-6 of the utterance peak, log-mel within 1e-4 of the float64 oracle), and within 2e-6 of the peak of that form's own output."""
    wav = synth.synth_waveforms(4) if kind == "white" else synth.synth_speechlike(4)
    outs = {}
    for compute in ("f32", "f32x3"):
        eng = Engine(model="none", compute=compute, max_batch=4)
        outs[compute] = eng.fbank(wav)
        eng.close()
    ref64 = o_fbank.melspectrogram(torch.from_numpy(wav).double()).numpy()
    peak = np.abs(ref64).max(axis=(1, 2), keepdims=True)
    e6 = float((np.abs(outs["f32x3"] - ref64) / peak).max())
    e1 = float((np.abs(outs["f32"] - ref64) / peak).max())
    lg = o_fbank.log_mean_norm(torch.from_numpy(outs["f32x3"]).double()).numpy()
    lr = o_fbank.log_mean_norm(torch.from_numpy(ref64)).numpy()
    e_log = float(np.abs(lg - lr).max())
    print(kind, "x6 rel-to-peak", e6, "fp32 MFMA rel-to-peak", e1, "x6 log-mel err", e_log, "x6 vs fp32 form", float((np.abs(outs["f32x3"] - outs["f32"]) / peak).max()))
    assert e6 <= 2e-6 and e_log <= 1e-4
    assert float((np.abs(outs["f32x3"] - outs["f32"]) / peak).max()) <= 2e-6


@pytest.mark.parametrize("L,B", [(32000, 5), (24000, 3), (10240 + 80 * 3, 3)])
def test_fused_front_end_of_bf16_handles(L, B):
    """Round 6: a bf16 handle's embed_wave runs waveform -> pre-emphasis -> DFT (window symmetry: two products of
    K = 101 on e = y[c+m] + y[c-m] and o = y[c+m] - y[c-m] instead of one of K = 200) -> power -> mel -> log -> minus the time mean -> the
    bf16 frame-major operand of blocks.0 in TWO launches (fbank_fused + its streaming normalisation) instead of fbank64 + prologue_stats +
    prologue_apply.  The operand (stage "input") against the float64 statement of the oracle — it is the bf16 rounding of values within
    ~1e-4 of it — and against the separate kernels (option fbank_unfused); utterance lengths whose last tile holds one frame tile, two, or
    a partial second one; white, speech-like (mel powers over several decades) and impulse-at-the-border waveforms; the embeddings of the
    two routes agree to bf16 round-off."""
    C = 512
    T = L // 80 + 1
    wav = np.concatenate([synth.synth_waveforms(1, L, seed=3), synth.synth_speechlike(B - 2, L), np.zeros((1, L), np.float32)])
    wav[-1, 0] = 1.0
    wav[-1, -1] = -0.5
    eng = Engine(model="ecapa", compute="bf16", channels=C, max_batch=B, samples=L)
    eng.load_state_dict(synth.synth_state_dict(synth.ecapa_param_spec(C=C), seed=3))
    eng.finalize()
    ref = o_fbank.log_mean_norm(o_fbank.melspectrogram(torch.from_numpy(wav).double())).numpy().transpose(0, 2, 1)      # (B, T, 80)
    out, xin, labels = {}, {}, {}
    for mode in (0, 1):
        eng.set_option("fbank_unfused", mode)
        eng.profile(True)
        out[mode] = eng.embed_wave(wav).copy()
        labels[mode] = set(eng.profile_results())
        eng.profile(False)
        xin[mode] = eng.get_stage("input").reshape(B, T, 80)
    again = eng.embed_wave(wav)
    assert np.array_equal(again, out[1])
    eng.set_option("fbank_unfused", 0)
    assert np.array_equal(eng.embed_wave(wav), out[0])                  # run to run: bitwise
    eng.close()
    assert "fbank_fused" in labels[0] and "prologue" not in labels[0] and "fbank" not in labels[0]
    assert "fbank" in labels[1] and "prologue" in labels[1] and "fbank_fused" not in labels[1]
    for mode in (0, 1):
        err = np.abs(xin[mode] - ref)
        bar = 2.0 ** -8 * np.abs(ref) + 1e-3           # half a bf16 ulp of the value + the split DFT's ~1e-4, with room for a flipped rounding
        print(f"L = {L}, {'separate' if mode else 'fused'} kernels: operand max |err| {err.max():.2e} (values up to {np.abs(ref).max():.1f}), "
              f"worst err / bar {float((err / bar).max()):.2f}")
        assert np.isfinite(xin[mode]).all() and float((err / bar).max()) <= 1.0
    d = np.abs(xin[0] - xin[1])
    assert float((d / (2.0 ** -7 * np.maximum(np.abs(xin[0]), np.abs(xin[1])) + 1e-3)).max()) <= 1.0      # at most one bf16 ulp apart
    assert float((d > 0).mean()) <= 0.05
    a, b = out[0], out[1]
    cos = np.sum(a * b, axis=1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))
    print(f"embeddings, fused vs separate front-end: cos >= {cos.min():.6f}")
    assert cos.min() >= 0.9999
