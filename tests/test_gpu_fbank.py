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
