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
    accepts_device_wave = True      # forward / embed_wave take CUDA tensors as raw device pointers (no host copy)
    ENGINE_CACHE = 3                # handles kept alive at once (one per input geometry)

    def __init__(self, spec, engine_kwargs, device=None, compute="f32", max_batch=256, seed=0, primary_samples=None):
        from .. import synth
        self._spec = OrderedDict((n, tuple(s)) for n, s in spec)
        self._sd = synth.synth_state_dict(spec, seed=seed)        # random init, like a fresh nn.Module
        self._engine_kwargs = dict(engine_kwargs)
        self._compute = compute
        self._max_batch = int(max_batch)
        self._device = _device_index(device)
        self._engines = OrderedDict()           # key -> Engine, most recently used last
        # the geometry (samples) that gets the full max_batch workspace: the configured crop length when the constructor knows
        # it (audio_spec), else the first length seen
        self._primary_cfg = int(primary_samples) if primary_samples else None
        self._primary = self._primary_cfg
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
        for eng in self._engines.values():
            eng.close()
        self._engines.clear()
        self._primary = self._primary_cfg

    @property
    def _engine(self):
        """the most recently used handle (tests read stage tensors / profiles from it)"""
        return next(reversed(self._engines.values())) if self._engines else None

    def _get_engine(self, samples, stream=None, batch=None):
        """One handle per input geometry.  The configured crop length (audio_spec: sentence_len * sample_rate; without it the
        FIRST geometry seen) gets the full `max_batch` workspace; any other length (whole-file evaluation, num_eval == 0: one forward per file, every file its
        own length) gets a workspace sized for the rows of that call, and at most ENGINE_CACHE handles stay alive — a
        one-minute file no longer allocates a max_batch x 10^4-frame workspace, and a repeated length is not rebuilt."""
        if self._primary is None:
            self._primary = samples
        mb = self._max_batch if samples == self._primary else max(1, min(self._max_batch, int(batch or 1)))
        key = (samples, self._device, self._compute, stream, mb)
        eng = self._engines.get(key)
        if eng is None:
            while len(self._engines) >= self.ENGINE_CACHE:
                victim = next((k for k in self._engines if k[0] != self._primary), next(iter(self._engines)))
                self._engines.pop(victim).close()
            # numeric status (SVHIP_ERR_NONFINITE / _RANGE): an fp16 handle that reports it may have produced something the reference would
            # not have (an fp16 overflow): raise; every other handle (bf16 has fp32's exponent range: a non-finite embedding there means a
            # non-finite input or weights) hands back what it computed, as the reference does, with a warning per batch
            kw = dict(self._engine_kwargs)
            kw.setdefault("on_numeric", "raise" if self._is_f16_handle() else "warn")
            eng = Engine(model=self.model_kind, compute=self._compute, max_batch=mb, samples=samples,
                         device=self._device, stream=stream, **kw)
            eng.load_state_dict(self._sd)
            eng.finalize()
            self._engines[key] = eng
        else:
            self._engines.move_to_end(key)
        return eng

    def _is_f16_handle(self):
        """the handle stores activations as IEEE half (RawNet2's 16-bit mode: compute 'f16', or 'half' on a RawNet2 model)"""
        return self._compute in ("f16", "fp16") or (self._compute == "half" and self.model_kind == "rawnet2")

    @staticmethod
    def _squeeze(out):
        """the reference ends forward with x.squeeze() (ECAPA_TDNN.py:500, RawNet2_custom.py:226)"""
        return out.squeeze() if _is_torch(out) else np.squeeze(out)

    def _batched(self, fn, x, max_batch=None):
        """run fn over chunks of at most max_batch rows and concatenate"""
        B = x.shape[0]
        mb = max_batch or self._max_batch
        if B <= mb:
            return fn(x)
        parts = [fn(x[i:i + mb]) for i in range(0, B, mb)]
        return torch.cat(parts, 0) if _is_torch(parts[0]) else np.concatenate(parts, 0)

    def __call__(self, x, *a, **k):
        return self.forward(x, *a, **k)
