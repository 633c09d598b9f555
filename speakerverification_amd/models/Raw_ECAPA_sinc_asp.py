"""Drop-in for the reference fusion plug-in ``models/Raw_ECAPA_sinc_asp.py`` (:22-52): the repo's
production model — ECAPA-TDNN (C = 512, 192-d) on the mel spectrogram of the waveform, concatenated
with RawNet2 (sinc / asp, nOut - 192 dims) on the raw waveform.

    model = MainModel(nOut=512, features='raw', n_mels=80, audio_spec={...})
    emb = model(wav)            # (B, 32000) -> (B, 512)

As in the reference the ECAPA branch inherits ``features`` from the config: with ``features: raw`` (what
every fusion YAML sets) it consumes the mel POWER without log / mean normalisation
(ECAPA_TDNN.py:473).  Both branches read the same waveform; the mel front-end and the ECAPA body run
as one fused library call.  State-dict keys: ``ECAPA_TDNN.*``, ``rawnet2v2.*`` (``compute_features.*``
buffers of nnAudio are accepted and ignored: the front-end tables are rebuilt from the config).
"""
from __future__ import annotations

import numpy as np

from ..engine import _is_torch
from . import ECAPA_TDNN as _ecapa
from . import RawNet2_custom as _rawnet2

try:
    import torch
except Exception:  # pragma: no cover
    torch = None


class Raw_ECAPA:
    accepts_device_wave = True      # both branches take CUDA tensors as raw device pointers

    def __init__(self, nOut=512, **kwargs):
        kw = dict(kwargs)
        kw.pop("channels", None)
        kw.pop("input_norm", None)
        self.ECAPA_TDNN = _ecapa.MainModel(nOut=192, channels=[512, 512, 512, 512, 1536], input_norm=False, **kw)
        self.rawnet2v2 = _rawnet2.MainModel(nOut=nOut - 192, front_proc="sinc", aggregate="asp", att_dim=128, **kw)
        self.training = False

    # nn.Module look-alikes --------------------------------------------------------------------------
    def to(self, device=None, *a, **k):
        self.ECAPA_TDNN.to(device)
        self.rawnet2v2.to(device)
        return self

    def eval(self):
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("training is outside the scope of the MI355X inference path")
        return self

    def parameters(self):
        yield from self.ECAPA_TDNN.parameters()
        yield from self.rawnet2v2.parameters()

    def state_dict(self):
        sd = {"ECAPA_TDNN." + k: v for k, v in self.ECAPA_TDNN.state_dict().items()}
        sd.update({"rawnet2v2." + k: v for k, v in self.rawnet2v2.state_dict().items()})
        return sd

    def load_state_dict(self, sd, strict=True):
        e = {k[len("ECAPA_TDNN."):]: v for k, v in sd.items() if k.startswith("ECAPA_TDNN.")}
        r = {k[len("rawnet2v2."):]: v for k, v in sd.items() if k.startswith("rawnet2v2.")}
        other = [k for k in sd if not k.startswith(("ECAPA_TDNN.", "rawnet2v2.", "compute_features."))]
        if strict and other:
            raise KeyError(f"unexpected keys {other[:4]}")
        m1 = self.ECAPA_TDNN.load_state_dict(e, strict=strict)
        m2 = self.rawnet2v2.load_state_dict(r, strict=strict)
        return m1, m2

    def load_blob(self, path):
        """the pair of branch blobs checkpoint.convert_checkpoint(..., model='Raw_ECAPA_sinc_asp') wrote for `path`"""
        from .. import checkpoint
        p_ecapa, p_rawnet2 = checkpoint.fusion_blob_paths(path)
        return self.ECAPA_TDNN.load_blob(p_ecapa), self.rawnet2v2.load_blob(p_rawnet2)

    def forward(self, x):
        out1 = self.ECAPA_TDNN.embed_wave(x)          # compute_features + ECAPA_TDNN (Raw_ECAPA_sinc_asp.py:41-44)
        out2 = self.rawnet2v2(x)                      # :48
        if _is_torch(out1):
            return torch.cat([out1, out2], dim=-1)    # :50
        return np.concatenate([out1, out2], axis=-1)

    __call__ = forward


def MainModel(nOut=512, **kwargs):
    return Raw_ECAPA(nOut=nOut, **kwargs)
