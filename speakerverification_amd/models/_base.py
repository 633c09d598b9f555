"""Shared plumbing of the MainModel shims: a torch.nn.Module-shaped object whose forward runs in
libsvhip.  Holds the reference-named state dict on the host; the device handle (Engine) is built
lazily for the input geometry of the first forward and rebuilt when weights or geometry change."""
from __future__ import annotations

from collections import OrderedDict

import numpy as np

from ..engine import Engine, _is_torch

try:
    import torch
except Exception:  # pragma: no cover
    torch = None


def _device_index(device) -> int:
    if device is None:
        return 0
    if torch is not None and isinstance(device, torch.device):
        return device.index or 0
    s = str(device)
    return int(s.split(":")[1]) if ":" in s else 0


class HipModule:
    """Mirror of the nn.Module surface the reference touches on ``__S__`` (src/model.py:73,119-121,
    180,209-210,235,332,712): forward/__call__, to, eval, train, parameters, state_dict, load_state_dict."""

    model_kind = "ecapa"

    def __init__(self, spec, engine_kwargs, device=None, compute="f32", max_batch=64, seed=0):
        from .. import synth
        self._spec = OrderedDict((n, tuple(s)) for n, s in spec)
        self._sd = synth.synth_state_dict(spec, seed=seed)        # random init, like a fresh nn.Module
        self._engine_kwargs = dict(engine_kwargs)
        self._compute = compute
        self._max_batch = int(max_batch)
        self._device = _device_index(device)
        self._engine = None
        self._engine_key = None
        self.training = False

    # ---- nn.Module look-alikes --------------------------------------------------------------------
    def to(self, device=None, *a, **k):
        idx = _device_index(device)
        if idx != self._device:
            self._device = idx
            self._drop_engine()
        return self

    def cuda(self, device=None):
        return self.to(device if device is not None else "cuda:0")

    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("training is outside the scope of the MI355X inference path")
        return self

    def parameters(self):
        for k, v in self._sd.items():
            if v.dtype != np.int64 and "running_" not in k:
                yield (torch.from_numpy(v) if torch is not None else v)

    def state_dict(self):
        if torch is None:
            return OrderedDict(self._sd)
        return OrderedDict((k, torch.from_numpy(np.asarray(v))) for k, v in self._sd.items())

    def load_state_dict(self, sd, strict=True):
        """Reference key names; with strict=False unknown / mis-shaped entries are skipped as
        ModelHandling.loadParameters does (src/model.py:730-742)."""
        missing = [k for k in self._spec if k not in sd and "num_batches_tracked" not in k]
        unexpected = [k for k in sd if k not in self._spec]
        if strict and (missing or unexpected):
            raise KeyError(f"state dict mismatch: missing {missing[:4]}..., unexpected {unexpected[:4]}...")
        for k, v in sd.items():
            if k not in self._spec:
                continue
            a = v.detach().cpu().numpy() if _is_torch(v) else np.asarray(v)
            if tuple(a.shape) != self._spec[k]:
                if strict:
                    raise ValueError(f"Wrong parameter length: {k}, model: {self._spec[k]}, loaded: {tuple(a.shape)}")
                continue
            self._sd[k] = a.astype(np.int64) if a.dtype == np.int64 else np.ascontiguousarray(a, dtype=np.float32)
        self._drop_engine()
        return missing, unexpected

    def load_blob(self, path):
        """Weights from a packed checkpoint blob (checkpoint.convert_checkpoint); tensors of other networks are ignored."""
        from .. import checkpoint, _lib
        mid, sd = checkpoint.read_blob(path)
        want = {"ecapa": _lib.MODEL_ECAPA, "rawnet2": _lib.MODEL_RAWNET2}.get(self.model_kind)
        if want is not None and mid != want:
            raise ValueError(f"{path} holds weights of model {mid}, this module is {self.model_kind}")
        return self.load_state_dict(sd, strict=False)

    # ---- engine management ----------------------------------------------------------------------------
    def _drop_engine(self):
        if self._engine is not None:
            self._engine.close()
        self._engine = None
        self._engine_key = None

    def _get_engine(self, samples, stream=None):
        key = (samples, self._device, self._compute, stream)
        if self._engine is None or self._engine_key != key:
            self._drop_engine()
            eng = Engine(model=self.model_kind, compute=self._compute, max_batch=self._max_batch, samples=samples,
                         device=self._device, stream=stream, **self._engine_kwargs)
            eng.load_state_dict(self._sd)
            eng.finalize()
            self._engine, self._engine_key = eng, key
        return self._engine

    @staticmethod
    def _squeeze(out):
        """the reference ends forward with x.squeeze() (ECAPA_TDNN.py:500, RawNet2_custom.py:226)"""
        return out.squeeze() if _is_torch(out) else np.squeeze(out)

    def _batched(self, fn, x):
        """run fn over chunks of at most max_batch rows and concatenate"""
        B = x.shape[0]
        if B <= self._max_batch:
            return fn(x)
        parts = [fn(x[i:i + self._max_batch]) for i in range(0, B, self._max_batch)]
        return torch.cat(parts, 0) if _is_torch(parts[0]) else np.concatenate(parts, 0)

    def __call__(self, x, *a, **k):
        return self.forward(x, *a, **k)
