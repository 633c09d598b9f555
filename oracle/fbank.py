"""Oracle (TEST INFRASTRUCTURE) — pre-emphasis + nnAudio-formulation mel spectrogram.

PARITY UNPINNED for F2–F3 (see ``oracle/__init__.py``): nnAudio (PyPI ``nnAudio``, listed bare at
reference ``src/requirements.txt:22``, no pinned version, absent from ``/root/reference`` and from this
image) owns the STFT and mel arithmetic.  Restated here from nnAudio 0.3.x's published algorithm,
anchored on the reference's call site ``src/models/FeatureExtraction/feature.py:66-94`` whose
defaults (sr=8000, n_fft=512, win_length=200, hop_length=80, window='hamming', fmin=0, fmax=None,
n_mels=80, pre_emphasis=True) are never overridden by any YAML.

``PreEmphasis`` (F1) *is* pinned: ``src/utils.py:53-71`` is imported by ``oracle/make_golden.py``.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F
from scipy.signal import get_window

FBANK_DEFAULTS = dict(sr=8000, n_fft=512, win_length=200, n_mels=80, hop_length=80,
                      window="hamming", fmin=0.0, fmax=None)      # feature.py:66-71


def pre_emphasis(x: torch.Tensor, coef: float = 0.97) -> torch.Tensor:
    """reference src/utils.py:63-71: reflect-pad one sample on the left, 2-tap conv [-coef, 1]."""
    assert x.dim() == 2
    xp = F.pad(x.unsqueeze(1), (1, 0), mode="reflect")
    flt = torch.tensor([[[-coef, 1.0]]], dtype=torch.float32).to(x.dtype)   # FloatTensor in the reference
    return F.conv1d(xp, flt).squeeze(1)


def window_taps(window="hamming", win_length=200, n_fft=512):
    """nnAudio ``create_fourier_kernels``: scipy periodic window, centred zero-pad to n_fft
    (``pad_center``: lpad = (n_fft - win_length)//2 = 156).  Returns (float32 window[n_fft], lpad)."""
    w = get_window(window, int(win_length), fftbins=True)
    lpad = (n_fft - win_length) // 2
    full = np.zeros(n_fft, dtype=np.float64)
    full[lpad:lpad + win_length] = w
    return full.astype(np.float32), lpad


def fourier_kernels(n_fft=512, win_length=200, window="hamming"):
    """nnAudio STFT kernels, freq_scale='no': wsin/wcos[k, s] = sin/cos(2*pi*k*s/n_fft) (float64,
    cast to float32) times the float32 window mask.  Shapes (n_fft//2+1, n_fft) float32."""
    s = np.arange(0, n_fft, 1.0)
    k = np.arange(n_fft // 2 + 1)[:, None]
    wsin = np.sin(2 * np.pi * k * s / n_fft).astype(np.float32)
    wcos = np.cos(2 * np.pi * k * s / n_fft).astype(np.float32)
    wmask, _ = window_taps(window, win_length, n_fft)
    return wsin * wmask[None, :], wcos * wmask[None, :]


def _hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)


def mel_basis(sr=8000, n_fft=512, n_mels=80, fmin=0.0, fmax=None):
    """nnAudio ``get_mel`` == librosa 0.7 ``filters.mel(htk=False, norm=1)`` (Slaney scale, area norm).
    Returns (n_mels, n_fft//2+1) float32."""
    if fmax is None:
        fmax = float(sr) / 2
    n_bins = 1 + n_fft // 2
    weights = np.zeros((n_mels, n_bins), dtype=np.float32)
    fftfreqs = np.linspace(0, float(sr) / 2, n_bins, endpoint=True)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis]          # in place on float32, as librosa does
    return weights


def power_spectrogram(y: torch.Tensor, n_fft=512, win_length=200, hop_length=80, window="hamming"):
    """nnAudio STFT(center=True, pad_mode='reflect', output_format='Magnitude') ** 2.0:
    ReflectionPad1d(n_fft//2); two conv1d's (stride hop) with the windowed sin / cos kernels;
    sqrt(re^2 + im^2) (no eps, non-trainable) then ``** 2.0``.  (B, L) -> (B, n_fft//2+1, T)."""
    wsin, wcos = fourier_kernels(n_fft, win_length, window)
    wsin = torch.from_numpy(wsin).to(y.dtype).unsqueeze(1)
    wcos = torch.from_numpy(wcos).to(y.dtype).unsqueeze(1)
    yp = F.pad(y.unsqueeze(1), (n_fft // 2, n_fft // 2), mode="reflect")
    im = F.conv1d(yp, wsin, stride=hop_length)
    re = F.conv1d(yp, wcos, stride=hop_length)
    mag = torch.sqrt(re.pow(2) + im.pow(2))
    return mag ** 2.0


def melspectrogram(x: torch.Tensor, pre_emph=True, **kw) -> torch.Tensor:
    """feature.py:66-94 with lib='nnaudio': Sequential(PreEmphasis, MelSpectrogram).
    (B, L) waveform -> (B, n_mels, T) mel *power* (no log; the log lives in the model, ECAPA_TDNN.py:473-476)."""
    p = dict(FBANK_DEFAULTS)
    p.update({k: v for k, v in kw.items() if k in p})
    y = pre_emphasis(x) if pre_emph else x
    spec = power_spectrogram(y, p["n_fft"], p["win_length"], p["hop_length"], p["window"])
    mb = torch.from_numpy(mel_basis(p["sr"], p["n_fft"], p["n_mels"], p["fmin"], p["fmax"])).to(x.dtype)
    return torch.matmul(mb, spec)


def log_mean_norm(mel: torch.Tensor) -> torch.Tensor:
    """ECAPA_TDNN.py:473-476: x = log(x + 1e-6); x = x - mean_t(x)."""
    x = (mel + 1e-6).log()
    return x - torch.mean(x, dim=-1, keepdim=True)
