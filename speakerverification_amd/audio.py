"""Eval-mode audio loading / cropping — host-side mirror of the reference's
``src/processing/audio_loader.py:53-152`` (``loadWAV``) and ``wav_conversion.py:35-41``.

File decoding is I/O, not the hot path: WAV files are read with ``soundfile`` when it is installed
(as the reference does) and with ``scipy.io.wavfile`` otherwise (PCM is scaled to [-1, 1) like
``sf.read(dtype='float32')``).
"""
from __future__ import annotations

from pathlib import Path

import numpy as np

DEFAULT_AUDIO_SPEC = {"sample_rate": 8000, "channels": 1, "sentence_len": 2.0, "win_len": 0.025, "hop_len": 0.01}


def read_wav(path):
    try:
        import soundfile as sf
        audio, sr = sf.read(str(path), dtype="float32", always_2d=False)
        return audio, sr
    except ImportError:
        from scipy.io import wavfile
        sr, a = wavfile.read(str(path))
        if np.issubdtype(a.dtype, np.integer):
            info = np.iinfo(a.dtype)
            a = a.astype(np.float32) / float(max(info.max, -info.min))
        return a.astype(np.float32), sr


def read_pcm16(path, sample_rate=None):
    """16-bit mono PCM WAV -> int16 samples (what svhip_crop_pcm16 consumes: scaled by 1/32768 on the device, exactly
    soundfile's float32 conversion), or None when the file is anything else (the caller then takes the host path)."""
    from scipy.io import wavfile
    try:
        sr, a = wavfile.read(str(path), mmap=False)
    except Exception:
        return None
    if a.dtype != np.int16 or a.ndim != 1 or a.size == 0:
        return None
    return a


def normalize_audio_amp(signal):
    """wav_conversion.py:35-41"""
    if np.issubdtype(signal.dtype, np.integer):
        info = np.iinfo(signal.dtype)
        return signal / max(info.max, -info.min)
    return signal / max(signal.max(), -signal.min())


def loadWAV(audio_source, audio_spec=None, evalmode=True, num_eval=10, augment=False, augment_options=None,
            random_chunk=False, load_all=False, dtype=np.float32, **kwargs):
    """Eval-mode subset of the reference loadWAV: returns (num_eval, max_audio) crops."""
    if not evalmode or augment:
        raise NotImplementedError("training-time loading / augmentation is outside the inference hot path")
    audio_spec = audio_spec or DEFAULT_AUDIO_SPEC
    if isinstance(audio_source, (str, Path)):
        audio, sr = read_wav(audio_source)
    elif isinstance(audio_source, np.ndarray):
        audio = normalize_audio_amp(audio_source)
    else:
        raise TypeError("Invalid format of audio source, available: str, ndarray")
    if load_all:
        return np.expand_dims(audio, 0)
    max_audio = int(audio_spec["sentence_len"] * audio_spec["sample_rate"])
    audiosize = audio.shape[0]
    if audiosize <= max_audio:
        audio = np.pad(audio, (0, max_audio - audiosize + 1), "wrap")
        audiosize = audio.shape[0]
    if num_eval == 0:
        return np.stack([audio], axis=0).astype(dtype)
    starts = np.linspace(0, audiosize - max_audio, num=num_eval)
    return np.stack([audio[int(s):int(s) + max_audio] for s in starts], axis=0).astype(dtype)
