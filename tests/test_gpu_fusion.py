"""GPU parity: the Raw_ECAPA_sinc_asp fusion model vs the output of the REFERENCE's own
module (tests/golden/fusion_raw_ecapa.npz; nnAudio front-end replaced by the oracle's restatement)."""
import os

import numpy as np
import pytest

from speakerverification_amd import synth
from speakerverification_amd.models import Raw_ECAPA_sinc_asp

pytestmark = pytest.mark.gpu

KW = dict(n_mels=80, augment=False, augment_options={"augment_chain": []}, features="raw",
          audio_spec=dict(sample_rate=16000, sentence_len=2.0, win_len=0.025, hop_len=0.01, channels=1))


def test_fusion_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "fusion_raw_ecapa.npz"))
    m = Raw_ECAPA_sinc_asp.MainModel(nOut=512, **KW)
    sd = {"ECAPA_TDNN." + k: v for k, v in synth.synth_state_dict(synth.ecapa_param_spec(C=512), seed=1).items()}
    sd.update({"rawnet2v2." + k: v for k, v in synth.synth_state_dict(synth.rawnet2_param_spec(nOut=320), seed=1).items()})
    sd["compute_features.0.flipped_filter"] = np.array([[[-0.97, 1.0]]], np.float32)     # reference checkpoints carry these
    m.load_state_dict(sd)
    assert len(m.state_dict()) == 231 + 147
    x = synth.synth_waveforms(2, 32000, seed=20220829)
    out = m(x)
    ref = g["out"]
    assert out.shape == ref.shape == (2, 512)
    err = float(np.abs(out - ref).max())
    assert err <= 1e-4 * float(np.abs(ref).max()), err
    on = out / np.linalg.norm(out, axis=1, keepdims=True)
    rn = ref / np.linalg.norm(ref, axis=1, keepdims=True)
    assert float(np.abs(on - rn).max()) <= 1e-4
    assert m(x[:1]).shape == (512,)


def test_fusion_blob_and_whole_file_engine_cache(tmp_path):
    """the production model through the blob path (one blob per branch) equals the state-dict path bit for bit; whole-file
    evaluation (a new length per call) keeps the fixed-length handle and at most ENGINE_CACHE handles alive."""
    from speakerverification_amd import checkpoint
    from speakerverification_amd.models import ECAPA_TDNN
    sd = {"__S__.ECAPA_TDNN." + k: v for k, v in synth.synth_state_dict(synth.ecapa_param_spec(C=512), seed=1).items()}
    sd.update({"__S__.rawnet2v2." + k: v for k, v in synth.synth_state_dict(synth.rawnet2_param_spec(nOut=320), seed=1).items()})
    dst = tmp_path / "fusion.svhip"
    checkpoint.convert_checkpoint(sd, dst, "Raw_ECAPA_sinc_asp")
    a = Raw_ECAPA_sinc_asp.MainModel(nOut=512, embed_batch=4, **KW)
    a.load_state_dict({k[len("__S__."):]: v for k, v in sd.items()})
    b = Raw_ECAPA_sinc_asp.MainModel(nOut=512, embed_batch=4, **KW)
    b.load_blob(dst)
    x = synth.synth_waveforms(3, 32000, seed=7)
    assert np.array_equal(a(x), b(x))
    # engine cache: fixed-length crops first, then three whole files of different lengths, then the crops again
    m = ECAPA_TDNN.MainModel(nOut=192, channels=[64] * 4 + [192], embed_batch=8)
    e0 = m.embed_wave(synth.synth_waveforms(8, 32000, seed=1))
    primary = m._engine
    for L in (40000, 48000, 56000, 64000):
        out = m.embed_wave(synth.synth_waveforms(1, L, seed=2))
        assert out.shape == (192,) and m._engine.max_batch == 1         # a workspace for ONE utterance of that length
        assert len(m._engines) <= m.ENGINE_CACHE
    e1 = m.embed_wave(synth.synth_waveforms(8, 32000, seed=1))
    assert m._engine is primary and np.array_equal(e0, e1)              # the fixed-length handle survived and was not rebuilt


def test_fusion_device_batch_runs_both_branches_concurrently():
    """A CUDA-tensor batch takes the two-stream path of Raw_ECAPA.forward (each branch on its handle's own stream): same
    numbers as the host path, bit for bit (same kernels, same order inside each branch)."""
    import torch
    m = Raw_ECAPA_sinc_asp.MainModel(nOut=512, embed_batch=8, **KW)
    sd = {"ECAPA_TDNN." + k: v for k, v in synth.synth_state_dict(synth.ecapa_param_spec(C=512), seed=1).items()}
    sd.update({"rawnet2v2." + k: v for k, v in synth.synth_state_dict(synth.rawnet2_param_spec(nOut=320), seed=1).items()})
    m.load_state_dict(sd)
    x = synth.synth_waveforms(5, 32000, seed=3)
    host = m(x)
    dev = m(torch.from_numpy(x).cuda())
    assert dev.is_cuda and tuple(dev.shape) == (5, 512)
    assert np.array_equal(dev.cpu().numpy(), host)
    big = m(torch.from_numpy(synth.synth_waveforms(11, 32000, seed=4)).cuda())        # more rows than embed_batch: chunked path
    assert tuple(big.shape) == (11, 512) and bool(torch.isfinite(big).all())
