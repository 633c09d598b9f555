"""Engine — thin Python owner of one svhip handle (one HIP device, one stream, one model).

Host-side plumbing only: every array that crosses into the library is passed as a raw pointer
(numpy -> host pointer, torch CUDA tensor -> device pointer).  No computation happens here.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np

from . import _lib

try:  # torch is plumbing (device memory / streams / torch.distributed); numpy-only use works without it
    import torch
except Exception:  # pragma: no cover
    torch = None


def _is_torch(x) -> bool:
    return torch is not None and isinstance(x, torch.Tensor)


# bytes that crossed PCIe through this module since import (host buffers handed to / filled by the library); tests use it
# to assert WHAT travels (e.g. int16 PCM instead of fp32 crops), bench.py to report PCIe-inclusive rates
TRANSFER_STATS = {"h2d_bytes": 0, "d2h_bytes": 0}


def _count(inputs=(), outputs=()):
    for b in inputs:
        if not b.device:
            TRANSFER_STATS["h2d_bytes"] += b.nbytes
    for b in outputs:
        if not b.device:
            TRANSFER_STATS["d2h_bytes"] += b.nbytes


def to_device(a, device=0):
    """host array -> CUDA tensor (torch owns the memory: plumbing), counted as PCIe traffic"""
    t = torch.from_numpy(np.ascontiguousarray(a)).to(f"cuda:{device}")
    TRANSFER_STATS["h2d_bytes"] += t.numel() * t.element_size()
    return t


def to_host(t):
    """CUDA tensor -> numpy, counted as PCIe traffic (numpy arrays pass through)"""
    if _is_torch(t):
        if t.is_cuda:
            TRANSFER_STATS["d2h_bytes"] += t.numel() * t.element_size()
        return t.detach().cpu().numpy()
    return np.asarray(t)


class _Buf:
    """Pointer view of a numpy array (host) or a torch tensor (host or device)."""

    def __init__(self, x, dtype, writable=False):
        self.keep = x
        if _is_torch(x):
            tdt = {np.float32: torch.float32, np.int32: torch.int32, np.int16: torch.int16, np.int64: torch.int64}[dtype]
            if x.dtype != tdt or not x.is_contiguous():
                if writable:
                    raise ValueError("output tensors must be contiguous and of the right dtype")
                x = x.to(tdt).contiguous()
                self.keep = x
            self.device = x.is_cuda
            self.ptr = x.data_ptr()
            self.nbytes = x.numel() * x.element_size()
        else:
            a = np.asarray(x)
            if a.dtype != dtype or not a.flags["C_CONTIGUOUS"]:
                if writable:
                    raise ValueError("output arrays must be C-contiguous and of the right dtype")
                a = np.ascontiguousarray(a, dtype=dtype)
            self.keep = a
            self.device = False
            self.ptr = a.ctypes.data
            self.nbytes = a.nbytes


class Engine:
    def __init__(self, model="ecapa", compute="f32", channels=1024, n_mels=80, embed_dim=192, max_batch=8,
                 samples=32000, log_input=True, input_norm=False, device=0, stream=None,
                 sr=8000, n_fft=512, win_length=200, hop_length=80, fmin=0.0, fmax=None, pre_emphasis=True, on_numeric="raise"):
        """``on_numeric``: what a forward does when the library reports SVHIP_ERR_NONFINITE / SVHIP_ERR_RANGE (the outputs were written,
        but an fp16 activation overflowed / the input left an f32x3 handle's range / the input was not finite): "raise"
        (_lib.SvhipNumericError, default), "warn" (RuntimeWarning; the outputs are returned as computed, as the reference would) or "ignore"."""
        if on_numeric not in ("raise", "warn", "ignore"):
            raise ValueError("on_numeric must be 'raise', 'warn' or 'ignore'")
        self.on_numeric = on_numeric
        self.lib = _lib.load()
        cfg = _lib.default_config()
        cfg.model = {"ecapa": _lib.MODEL_ECAPA, "rawnet2": _lib.MODEL_RAWNET2, "none": _lib.MODEL_NONE}[model]
        cfg.compute = {"f32": _lib.F32, "fp32": _lib.F32, "bf16": _lib.BF16, "f32x3": _lib.F32X3, "bf16x3": _lib.F32X3,
                       "f16": _lib.F16, "fp16": _lib.F16}[compute]
        cfg.device = int(device)
        cfg.channels = int(channels)
        cfg.n_mels = int(n_mels)
        cfg.embed_dim = int(embed_dim)
        cfg.max_batch = int(max_batch)
        cfg.samples = int(samples)
        cfg.log_input = int(bool(log_input))
        cfg.input_norm = int(bool(input_norm))
        cfg.fb_sr, cfg.n_fft, cfg.win_length, cfg.hop_length = int(sr), int(n_fft), int(win_length), int(hop_length)
        cfg.fmin = float(fmin)
        cfg.fmax = -1.0 if fmax is None else float(fmax)
        cfg.preemph = 0.97 if pre_emphasis is True else (-1.0 if pre_emphasis in (False, None) else float(pre_emphasis))
        cfg.stream = C.c_void_p(int(stream)) if stream else None
        self.cfg = cfg
        self.model = model
        self.compute = {_lib.BF16: "bf16", _lib.F32X3: "f32x3", _lib.F16: "f16"}.get(cfg.compute, "f32")
        self.max_batch = cfg.max_batch
        self.samples = cfg.samples
        self.frames = cfg.samples // cfg.hop_length + 1
        self.n_mels = cfg.n_mels
        self.embed_dim = cfg.embed_dim
        self.device = cfg.device
        h = C.c_void_p()
        rc = self.lib.svhip_create(C.byref(cfg), C.byref(h))
        if rc != _lib.OK:
            msg = self.lib.svhip_last_error(None)
            raise _lib.SvhipError(rc, msg.decode() if msg else "?")
        self.h = h
        self._stream = int(stream) if stream else None

    def _order_after_torch(self, *bufs, async_=False, ordered=False):
        """Device tensors handed in by torch may still be in flight on torch's current stream.  When the
        handle runs on that same stream the order is already right; otherwise wait for torch's stream
        on the host before the library touches the data (and refuse async calls, which would race)."""
        if ordered or torch is None or not any(b.device for b in bufs):
            return                                  # (ordered: the caller has already waited for torch's stream)
        cur = torch.cuda.current_stream(self.device)
        if self._stream is not None and cur.cuda_stream == self._stream:
            return
        if async_:
            raise ValueError("async_ calls need the Engine to be created on torch's current stream")
        cur.synchronize()

    # ---- lifetime ------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "h", None):
            self.lib.svhip_destroy(self.h)
            self.h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        _lib.check(self.h, rc, self.on_numeric)

    def synchronize(self):
        """waits for the handle's stream; reports (per on_numeric) and clears the numeric status of the async calls it waited for"""
        self._ck(self.lib.svhip_synchronize(self.h))

    def numeric_status(self, reset=True):
        """0, _lib.ERR_NONFINITE or _lib.ERR_RANGE for the forwards since the last reset (waits for the stream; never raises)"""
        return int(self.lib.svhip_numeric_status(self.h, 1 if reset else 0))

    # ---- weights ---------------------------------------------------------------------------------------
    def load_state_dict(self, sd: Dict[str, object], strict=True, show_error=False):
        """Name-matched copy of a reference ``__S__`` state dict (src/model.py:718-746 semantics:
        with strict=False unknown / mis-shaped tensors are skipped, optionally printed)."""
        skipped = []
        for name, v in sd.items():
            if _is_torch(v):
                v = v.detach().cpu().numpy()
            a = np.asarray(v)
            shp = a.shape                       # (ascontiguousarray would promote 0-d to 1-d)
            if a.dtype == np.int64:
                dt = _lib.I64
                a = np.ascontiguousarray(a)
            else:
                a = np.ascontiguousarray(a, dtype=np.float32)
                dt = _lib.F32
            shape = (C.c_int64 * max(1, len(shp)))(*shp)
            rc = self.lib.svhip_load_tensor(self.h, name.encode(), a.ctypes.data, shape, len(shp), dt)
            if rc != _lib.OK:
                if strict:
                    self._ck(rc)
                msg = self.lib.svhip_last_error(self.h).decode()
                skipped.append(name)
                if show_error:
                    print(msg)
        return skipped

    def finalize(self):
        self._ck(self.lib.svhip_finalize_weights(self.h))

    def load_blob(self, path):
        """mmap a packed checkpoint blob (speakerverification_amd.checkpoint) and finalize: svhip_load_blob."""
        self._ck(self.lib.svhip_load_blob(self.h, str(path).encode()))

    # ---- forward ------------------------------------------------------------------------------------------
    def _out(self, like, shape):
        if _is_torch(like) and like.is_cuda:
            return torch.empty(shape, dtype=torch.float32, device=like.device)
        return np.empty(shape, dtype=np.float32)

    @staticmethod
    def _flags(i: _Buf, o: _Buf, async_=False):
        f = (_lib.IN_DEVICE if i.device else 0) | (_lib.OUT_DEVICE if o.device else 0)
        if async_:
            f |= _lib.ASYNC
        return f

    def fbank(self, wav, out=None, async_=False):
        """(B, L) fp32 waveforms -> (B, n_mels, T) mel power (numpy in -> numpy out, CUDA tensor in -> CUDA tensor out)."""
        B, L = wav.shape
        if out is None:
            out = self._out(wav, (B, self.n_mels, self.frames))
        i, o = _Buf(wav, np.float32), _Buf(out, np.float32, writable=True)
        self._order_after_torch(i, o, async_=async_)
        _count([i], [o])
        self._ck(self.lib.svhip_fbank(self.h, i.ptr, B, L, o.ptr, self._flags(i, o, async_)))
        return out

    def embed_features(self, feat, out=None, async_=False):
        B, nm, T = feat.shape
        if nm != self.n_mels:
            raise ValueError(f"expected {self.n_mels} feature channels, got {nm}")
        if out is None:
            out = self._out(feat, (B, self.embed_dim))
        i, o = _Buf(feat, np.float32), _Buf(out, np.float32, writable=True)
        self._order_after_torch(i, o, async_=async_)
        _count([i], [o])
        self._ck(self.lib.svhip_embed_features(self.h, i.ptr, B, T, o.ptr, self._flags(i, o, async_)))
        return out

    def embed_wave(self, wav, out=None, async_=False, ordered=False):
        """``ordered=True``: the caller guarantees that the device buffers are complete and stay alive (it has synchronised
        torch's stream and will call synchronize() on this handle): lets an async call run on the handle's OWN stream."""
        B, L = wav.shape
        if out is None:
            out = self._out(wav, (B, self.embed_dim))
        i, o = _Buf(wav, np.float32), _Buf(out, np.float32, writable=True)
        self._order_after_torch(i, o, async_=async_, ordered=ordered)
        _count([i], [o])
        self._ck(self.lib.svhip_embed_wave(self.h, i.ptr, B, L, o.ptr, self._flags(i, o, async_)))
        return out

    def crop_pcm16(self, pcm_list, num_eval, L=32000, out=None, async_=False):
        """list of 1-D int16 arrays (decoded files) -> (len(list) * num_eval, L) fp32 eval-mode crops, cropped on
        the device (int16 travels over PCIe; reference semantics of loadWAV for 16-bit files).  With ``out`` a CUDA
        tensor the crops stay in HBM (the embed call then takes them as a device pointer: no fp32 crop crosses PCIe)."""
        lens = np.asarray([len(a) for a in pcm_list], dtype=np.int32)
        offs = np.zeros(len(pcm_list), dtype=np.int64)
        offs[1:] = np.cumsum(lens[:-1], dtype=np.int64)
        if len(pcm_list) == 1 and _is_torch(pcm_list[0]):          # one packed (pinned) int16 tensor holding the files back to back
            raise ValueError("pass packed PCM through crop_pcm16_packed")
        pcm = np.ascontiguousarray(np.concatenate([np.asarray(a, dtype=np.int16) for a in pcm_list]))
        return self.crop_pcm16_packed(pcm, offs, lens, num_eval, L, out=out, async_=async_)

    def crop_pcm16_packed(self, pcm, offs, lens, num_eval, L=32000, out=None, async_=False):
        """the files back to back in ONE int16 array (numpy, or a pinned torch tensor for truly asynchronous copies) with their
        offsets / lengths; async_=True (device `out` only) returns once the copy and the crop kernel are enqueued."""
        offs = np.ascontiguousarray(offs, dtype=np.int64)
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        p = _Buf(pcm, np.int16)
        if p.device:
            raise ValueError("packed PCM lives on the host (device PCM: call the C ABI with SVHIP_IN_DEVICE)")
        n_files = int(lens.shape[0])
        n = n_files * num_eval
        if out is None:
            out = np.empty((n, L), dtype=np.float32)
        elif tuple(out.shape) != (n, L):
            raise ValueError(f"out must be ({n}, {L})")
        o = _Buf(out, np.float32, writable=True)
        if async_ and not o.device:
            raise ValueError("async_ needs a device output tensor")
        self._order_after_torch(o, async_=async_)
        TRANSFER_STATS["h2d_bytes"] += p.nbytes + offs.nbytes + lens.nbytes
        _count([], [o])
        flags = (_lib.OUT_DEVICE if o.device else 0) | (_lib.ASYNC if async_ else 0)
        self._ck(self.lib.svhip_crop_pcm16(self.h, p.ptr, p.nbytes // 2, offs.ctypes.data, lens.ctypes.data, n_files,
                                           int(num_eval), int(L), o.ptr, flags))
        return out

    def synth_waveforms(self, seed, first_utt, B, L=None, out=None, async_=False):
        """utterances [first_utt, first_utt + B) of the counter-based synthetic stream ``seed`` (svhip_synth_waveforms)."""
        L = int(L or self.samples)
        if out is None:
            out = np.empty((B, L), dtype=np.float32)
        o = _Buf(out, np.float32, writable=True)
        self._order_after_torch(o, async_=async_)
        _count([], [o])
        flags = (_lib.OUT_DEVICE if o.device else 0) | (_lib.ASYNC if async_ else 0)
        self._ck(self.lib.svhip_synth_waveforms(self.h, int(seed), int(first_utt), int(B), L, o.ptr, flags))
        return out

    # ---- multi-GPU exchange (RCCL under the C ABI) -------------------------------------------------------------
    @staticmethod
    def comm_unique_id() -> bytes:
        lib = _lib.load()
        buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
        rc = lib.svhip_comm_unique_id(buf)
        if rc != _lib.OK:
            raise _lib.SvhipError(rc, (lib.svhip_comm_last_error() or b"?").decode())
        return buf.raw

    def comm_init(self, id_bytes: bytes, rank: int, world: int):
        if len(id_bytes) != _lib.COMM_ID_BYTES:
            raise ValueError("the RCCL unique id is %d bytes" % _lib.COMM_ID_BYTES)
        self._ck(self.lib.svhip_comm_init(self.h, C.create_string_buffer(id_bytes, _lib.COMM_ID_BYTES), int(rank), int(world)))
        self.comm_rank, self.comm_world = int(rank), int(world)

    def comm_rank_world(self):
        """(rank, world) as the library's RCCL communicator reports them (svhip_comm_rank)"""
        r, w = C.c_int32(), C.c_int32()
        self._ck(self.lib.svhip_comm_rank(self.h, C.byref(r), C.byref(w)))
        return r.value, w.value

    def allgather_rows(self, local, out=None, async_=False):
        """(rows, D) fp32 block of every rank -> (world * rows, D) on every rank: ONE RCCL all-gather on the handle's stream."""
        rows, D = local.shape
        world = getattr(self, "comm_world", None)
        if world is None:
            raise RuntimeError("comm_init has not been called on this Engine")
        if out is None:
            out = self._out(local, (world * rows, D))
        i, o = _Buf(local, np.float32), _Buf(out, np.float32, writable=True)
        if tuple(out.shape) != (world * rows, D):
            raise ValueError(f"out must be ({world * rows}, {D})")
        self._order_after_torch(i, o, async_=async_)
        _count([i], [o])
        self._ck(self.lib.svhip_allgather_rows(self.h, i.ptr, rows, D, o.ptr, self._flags(i, o, async_)))
        return out

    # ---- scoring ---------------------------------------------------------------------------------------------
    def l2norm_(self, E):
        N, D = E.shape
        b = _Buf(E, np.float32, writable=True)
        self._order_after_torch(b)
        self._ck(self.lib.svhip_l2norm(self.h, b.ptr, N, D, _lib.IN_DEVICE | _lib.OUT_DEVICE if b.device else 0))
        _count([b], [b])
        return E

    def _same_space(self, *bufs):
        dev = {b.device for b in bufs}
        if len(dev) != 1:
            raise ValueError("all arrays of one scoring call must live in the same memory space")
        self._order_after_torch(*bufs)
        return dev.pop()

    def score_pairs(self, E, ia, ib, out=None):
        N, D = E.shape
        P = int(ia.shape[0])
        if out is None:
            out = self._out(E, (P,))
        e, a, b, o = _Buf(E, np.float32), _Buf(ia, np.int32), _Buf(ib, np.int32), _Buf(out, np.float32, writable=True)
        dev = self._same_space(e, a, b, o)
        self._ck(self.lib.svhip_score_pairs(self.h, e.ptr, N, D, a.ptr, b.ptr, P, o.ptr,
                                            (_lib.IN_DEVICE | _lib.OUT_DEVICE) if dev else 0))
        _count([e, a, b], [o])
        return out

    TRIAL_MODES = {"cosine": _lib.TRIAL_COSINE, "pnorm": _lib.TRIAL_PNORM, "pdist": _lib.TRIAL_PDIST}

    def score_trials(self, F, ia, ib, mode="cosine", out=None, p=2):
        """whole-trial scores over the crops of two files: F (n_files, n_crops, D), ia / ib file indices (svhip_score_trials);
        mode "pnorm" takes the reference's `p` (utils.py:167: any float, +-inf, 0)"""
        n_files, n_crops, D = F.shape
        P = int(ia.shape[0])
        if out is None:
            out = self._out(F, (P,))
        f, a, b, o = _Buf(F, np.float32), _Buf(ia, np.int32), _Buf(ib, np.int32), _Buf(out, np.float32, writable=True)
        dev = self._same_space(f, a, b, o)
        fl = (_lib.IN_DEVICE | _lib.OUT_DEVICE) if dev else 0
        if mode == "pnorm" and float(p) != 2.0:
            self._ck(self.lib.svhip_score_trials_pnorm(self.h, float(p), f.ptr, n_files, n_crops, D, a.ptr, b.ptr, P, o.ptr, fl))
        else:
            self._ck(self.lib.svhip_score_trials(self.h, self.TRIAL_MODES[mode], f.ptr, n_files, n_crops, D, a.ptr, b.ptr, P, o.ptr, fl))
        _count([f, a, b], [o])
        return out

    def mean_crops(self, F, out=None):
        """(n_files, n_crops, D) -> (n_files, D) crop means"""
        n_files, n_crops, D = F.shape
        if out is None:
            out = self._out(F, (n_files, D))
        f, o = _Buf(F, np.float32), _Buf(out, np.float32, writable=True)
        dev = self._same_space(f, o)
        self._ck(self.lib.svhip_mean_crops(self.h, f.ptr, n_files, n_crops, D, o.ptr, (_lib.IN_DEVICE | _lib.OUT_DEVICE) if dev else 0))
        _count([f], [o])
        return out

    def score_matrix(self, A, B, out=None):
        Na, D = A.shape
        Nb = B.shape[0]
        if out is None:
            out = self._out(A, (Na, Nb))
        a, b, o = _Buf(A, np.float32), _Buf(B, np.float32), _Buf(out, np.float32, writable=True)
        dev = self._same_space(a, b, o)
        self._ck(self.lib.svhip_score_matrix(self.h, a.ptr, Na, b.ptr, Nb, D, o.ptr,
                                             (_lib.IN_DEVICE | _lib.OUT_DEVICE) if dev else 0))
        return out

    # ---- verification metrics (host arrays in / out; the trial list is small next to the embeddings) --------------------
    def _scores_labels(self, scores, labels):
        if _is_torch(scores):
            scores = scores.detach().cpu().numpy()
        if _is_torch(labels):
            labels = labels.detach().cpu().numpy()
        s = np.ascontiguousarray(np.asarray(scores, dtype=np.float32).ravel())
        l = np.ascontiguousarray(np.asarray(labels).ravel().astype(np.int32))
        if s.shape != l.shape or s.size == 0:
            raise ValueError("scores and labels must be non-empty and of equal length")
        return s, l

    def roc_points(self, scores, labels):
        """-> (fps, tps, thresholds) of sklearn's _binary_clf_curve (float64 counts, highest threshold first)."""
        s, l = self._scores_labels(scores, labels)
        P = s.size
        thr = np.empty(P, np.float32)
        fps, tps = np.empty(P, np.int64), np.empty(P, np.int64)
        n = C.c_int64()
        self._ck(self.lib.svhip_roc_points(self.h, s.ctypes.data, l.ctypes.data, P, C.byref(n), thr.ctypes.data,
                                           fps.ctypes.data, tps.ctypes.data, 0))
        k = n.value
        return fps[:k].astype(np.float64), tps[:k].astype(np.float64), thr[:k].copy()

    def error_rates(self, scores, labels):
        """ComputeErrorRates (utils.py:221-256) -> (fnrs, fprs float64, thresholds float32), ascending thresholds."""
        s, l = self._scores_labels(scores, labels)
        P = s.size
        fnrs, fprs, thr = np.empty(P, np.float64), np.empty(P, np.float64), np.empty(P, np.float32)
        self._ck(self.lib.svhip_error_rates(self.h, s.ctypes.data, l.ctypes.data, P, fnrs.ctypes.data, fprs.ctypes.data,
                                            thr.ctypes.data, 0))
        return fnrs, fprs, thr

    def min_dcf(self, scores, labels, p_target, c_miss, c_fa):
        """ComputeErrorRates + ComputeMinDcf (utils.py:221-275) -> (min_dcf, threshold)."""
        s, l = self._scores_labels(scores, labels)
        d, t = C.c_double(), C.c_float()
        self._ck(self.lib.svhip_min_dcf(self.h, s.ctypes.data, l.ctypes.data, s.size, float(p_target), float(c_miss),
                                        float(c_fa), C.byref(d), C.byref(t), 0))
        return d.value, t.value

    def asnorm_stats(self, E, cohort, top=200):
        N, D = E.shape
        K = cohort.shape[0]
        mu, sd = self._out(E, (N,)), self._out(E, (N,))
        e, c, m, s = _Buf(E, np.float32), _Buf(cohort, np.float32), _Buf(mu, np.float32, True), _Buf(sd, np.float32, True)
        dev = self._same_space(e, c, m, s)
        self._ck(self.lib.svhip_asnorm_stats(self.h, e.ptr, N, D, c.ptr, K, int(top), m.ptr, s.ptr,
                                             (_lib.IN_DEVICE | _lib.OUT_DEVICE) if dev else 0))
        _count([e, c], [m, s])
        return mu, sd

    @property
    def asnorm_last_fallback(self) -> int:
        """embeddings of the last asnorm_stats call that took the slab path after the fused kernel (-1: the whole call did)"""
        return int(self.lib.svhip_asnorm_last_fallback(self.h))

    @property
    def asnorm_last_refit(self):
        """(rows, passes): embeddings of the last asnorm_stats call that the fused kernel decided with a threshold re-derived from its own
        counts (non-Gaussian cohort scores), and the extra passes over them that took"""
        import ctypes
        n = ctypes.c_int32(0)
        rows = int(self.lib.svhip_asnorm_last_refit(self.h, ctypes.byref(n)))
        return rows, int(n.value)

    def asnorm_pairs(self, E, mu, sigma, ia, ib, out=None):
        N, D = E.shape
        P = int(ia.shape[0])
        if out is None:
            out = self._out(E, (P,))
        e, m, s = _Buf(E, np.float32), _Buf(mu, np.float32), _Buf(sigma, np.float32)
        a, b, o = _Buf(ia, np.int32), _Buf(ib, np.int32), _Buf(out, np.float32, writable=True)
        dev = self._same_space(e, m, s, a, b, o)
        self._ck(self.lib.svhip_asnorm_pairs(self.h, e.ptr, N, D, m.ptr, s.ptr, a.ptr, b.ptr, P, o.ptr,
                                             (_lib.IN_DEVICE | _lib.OUT_DEVICE) if dev else 0))
        _count([e, m, s, a, b], [o])
        return out

    # ---- introspection ------------------------------------------------------------------------------------------
    def get_stage(self, name: str) -> np.ndarray:
        n = C.c_int64()
        self._ck(self.lib.svhip_get_stage(self.h, name.encode(), None, C.byref(n)))
        out = np.empty(n.value, dtype=np.float32)
        self._ck(self.lib.svhip_get_stage(self.h, name.encode(), out.ctypes.data, C.byref(n)))
        return out

    def set_option(self, name: str, value: int):
        """Developer / test switch of this handle (``svhip_set_option``; the SVHIP_<NAME> environment variables are only the
        defaults a handle is created with)."""
        self._ck(self.lib.svhip_set_option(self.h, name.encode(), int(value)))

    def trim_scratch(self):
        """Free the scoring / metrics scratch the handle has grown (``svhip_trim_scratch``)."""
        self._ck(self.lib.svhip_trim_scratch(self.h))

    def profile(self, on=True, only=None):
        """HIP events around every kernel launch (``only``: just the launches with that label — fewer events in the stream)."""
        self._ck(self.lib.svhip_profile_filter(self.h, only.encode() if only else None))
        self._ck(self.lib.svhip_profile_enable(self.h, int(on)))
        if on:
            self._ck(self.lib.svhip_profile_reset(self.h))

    def profile_results(self):
        res = {}
        idx = 0
        name = C.create_string_buffer(64)
        ms, n, fl = C.c_double(), C.c_int64(), C.c_double()
        while self.lib.svhip_profile_get(self.h, idx, name, 64, C.byref(ms), C.byref(n), C.byref(fl)) == _lib.OK:
            res[name.value.decode()] = {"ms": ms.value, "launches": n.value, "flops": fl.value}
            idx += 1
        return res

    @property
    def flops_per_utterance(self) -> float:
        return float(self.lib.svhip_workload_flops(self.h))
