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
        if _is_torch(x) and x.is_cuda and x.ndim == 2 and x.shape[1] == self.rawnet2v2.nb_samp:
            # device-resident batch: the two branches run CONCURRENTLY, each on its handle's own stream (RawNet2's small late
            # kernels beside ECAPA's GEMMs: 61 k instead of 55 k utt/s at B = 256)
            e1 = self.ECAPA_TDNN._get_engine(x.shape[1], batch=x.shape[0])
            e2 = self.rawnet2v2._get_engine(x.shape[1])
            if x.shape[0] <= min(e1.max_batch, e2.max_batch) and x.dtype == torch.float32 and x.is_contiguous():
                torch.cuda.current_stream(x.device).synchronize()          # x is complete before either handle reads it
                out = torch.empty((x.shape[0], e1.embed_dim + e2.embed_dim), device=x.device, dtype=torch.float32)
                o1 = torch.empty((x.shape[0], e1.embed_dim), device=x.device, dtype=torch.float32)
                o2 = torch.empty((x.shape[0], e2.embed_dim), device=x.device, dtype=torch.float32)
                e1.embed_wave(x, out=o1, async_=True, ordered=True)       # compute_features + ECAPA_TDNN (Raw_ECAPA_sinc_asp.py:41-44)
                e2.embed_wave(x, out=o2, async_=True, ordered=True)       # :48
                e1.synchronize()
                e2.synchronize()
                out[:, :e1.embed_dim] = o1                                 # torch.cat([out1, out2], dim=-1)   :50
                out[:, e1.embed_dim:] = o2
                return out.squeeze()
        out1 = self.ECAPA_TDNN.embed_wave(x)          # compute_features + ECAPA_TDNN (Raw_ECAPA_sinc_asp.py:41-44)
        out2 = self.rawnet2v2(x)                      # :48
        if _is_torch(out1):
            return torch.cat([out1, out2], dim=-1)    # :50
        return np.concatenate([out1, out2], axis=-1)

    __call__ = forward


def MainModel(nOut=512, **kwargs):
    return Raw_ECAPA(nOut=nOut, **kwargs)
