"""Drop-in for the reference ``models/FeatureExtraction/feature.py`` (melspectrogram factory :66-94).

    compute_features = melspectrogram(**kwargs).to(device)
    mel = compute_features(wav)       # (B, L) -> (B, n_mels, T) mel power

Only ``lib='nnaudio'`` (the reference default, the only one any config uses) is built.
"""
from __future__ import annotations

import numpy as np

from ...engine import Engine, _is_torch
from .._base import _device_index

try:
    import torch
except Exception:  # pragma: no cover
    torch = None


class MelSpectrogram:
    """Sequential(PreEmphasis(), nnAudio MelSpectrogram) as one HIP kernel."""

    def __init__(self, sr=8000, n_fft=512, win_length=200, n_mels=80, hop_length=80, window="hamming",
                 fmin=0.0, fmax=None, pre_emphasis=True, max_batch=64, device=None):
        if window != "hamming":
            raise NotImplementedError("only the hamming window of feature.py:68 is built")
        self.p = dict(sr=sr, n_fft=n_fft, win_length=win_length, n_mels=n_mels, hop_length=hop_length,
                      fmin=fmin, fmax=fmax, pre_emphasis=pre_emphasis)
        self.max_batch = max_batch
        self._device = _device_index(device)
        self._engine = None
        self._samples = None

    def to(self, device=None, *a, **k):
        idx = _device_index(device)
        if idx != self._device:
            self._device = idx
            self._engine = None
        return self

    def eval(self):
        return self

    def _get_engine(self, samples):
        if self._engine is None or self._samples != samples:
            if self._engine is not None:
                self._engine.close()
            self._engine = Engine(model="none", max_batch=self.max_batch, samples=samples, device=self._device, **self.p)
            self._samples = samples
        return self._engine

    def __call__(self, x):
        if x.ndim != 2:
            raise AssertionError(f"The number of dimensions of input tensor must be 2, got {tuple(x.shape)}")  # utils.py:65-66
        eng = self._get_engine(x.shape[1])
        B = x.shape[0]
        if B <= self.max_batch:
            return eng.fbank(x)
        parts = [eng.fbank(x[i:i + self.max_batch]) for i in range(0, B, self.max_batch)]
        return torch.cat(parts, 0) if _is_torch(parts[0]) else np.concatenate(parts, 0)

    forward = __call__


def melspectrogram(lib="nnaudio", sr=8000, n_fft=512, win_length=200, n_mels=80, hop_length=80, window="hamming",
                   fmin=0.0, fmax=None, verbose=False, pre_emphasis=True, **kwargs):
    if lib.lower() != "nnaudio":
        raise NotImplementedError("only lib='nnaudio' (the reference default) is built")
    return MelSpectrogram(sr=sr, n_fft=n_fft, win_length=win_length, n_mels=n_mels, hop_length=hop_length,
                          window=window, fmin=fmin, fmax=fmax, pre_emphasis=pre_emphasis,
                          device=kwargs.get("device"))


def mfcc(*a, **k):
    raise NotImplementedError("mfcc features are outside the hot path (no reference config uses them)")
