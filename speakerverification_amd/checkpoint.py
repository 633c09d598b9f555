"""Packed checkpoint blobs: reference checkpoints -> the archive libsvhip.so mmaps.

The reference keeps weights as torch pickles written by ``trainer.py:145-205`` (``<save_path>/model/*.model``, the
``state_dict()`` of ``WrappedModel(SpeakerEncoder)``: keys ``module.__S__.*`` for the embedding network, ``module.__L__.*``
for the loss head, ``module.compute_features.*`` for the front-end buffers) and reads them back with ``torch.load`` in
``ModelHandling.loadParameters`` (``model.py:718-746``).  ``convert_checkpoint`` does that read once, on the host, keeps
the ``__S__`` tensors under their bare reference names and writes a blob (``csrc/blob.hip`` documents the layout);
``Engine.load_blob`` / ``svhip_load_blob`` then need neither Python pickles nor torch.

    python -m speakerverification_amd.checkpoint SRC.model DST.svhip --model ECAPA_TDNN

The blob reader / writer is the C one (ctypes), so there is a single implementation of the format.
"""
from __future__ import annotations

import argparse
import ctypes as C
from collections import OrderedDict

import numpy as np

from . import _lib

MAGIC = b"SVHIPWB1"
_MODEL_IDS = {"ECAPA_TDNN": _lib.MODEL_ECAPA, "ecapa": _lib.MODEL_ECAPA, "RawNet2_custom": _lib.MODEL_RAWNET2,
              "rawnet2": _lib.MODEL_RAWNET2}


def model_id(model) -> int:
    if isinstance(model, int):
        return model
    try:
        return _MODEL_IDS[model]
    except KeyError:
        raise ValueError(f"unknown model {model!r}; expected one of {sorted(_MODEL_IDS)}") from None


def is_blob(path) -> bool:
    try:
        with open(path, "rb") as f:
            return f.read(8) == MAGIC
    except OSError:
        return False


def _blob_error(rc):
    msg = _lib.load().svhip_blob_last_error()
    return _lib.SvhipError(rc, msg.decode() if msg else "?")


def embedding_state_dict(state) -> "OrderedDict[str, np.ndarray]":
    """The ``__S__`` part of a reference checkpoint under bare parameter names (what ``__S__.state_dict()`` would give).

    Accepts a ``WrappedModel`` / ``SpeakerEncoder`` state dict (``module.__S__.x`` / ``__S__.x`` keys; everything else
    — loss head, front-end buffers — is dropped, as the eval path never reads it) or an already bare state dict.
    """
    out = OrderedDict()
    keys = list(state.keys())
    has_s = any("__S__." in k for k in keys)
    for k in keys:
        v = state[k]
        if has_s:
            i = k.find("__S__.")
            if i < 0:
                continue
            name = k[i + len("__S__."):]
        else:
            name = k[len("module."):] if k.startswith("module.") else k
        a = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
        out[name] = a
    return out


def write_blob(path, model, state) -> None:
    """Write ``state`` (name -> array; fp32 or int64 tensors of rank <= 4) as a weight blob for ``model``."""
    lib = _lib.load()
    names, arrays = [], []
    for k, v in state.items():
        shape = tuple(np.shape(v))
        a = np.asarray(v)
        if a.dtype == np.int64 or a.dtype == np.int32:
            a = np.ascontiguousarray(a, dtype=np.int64)
        else:
            a = np.ascontiguousarray(a, dtype=np.float32)
        if len(shape) > 4:
            raise ValueError(f"{k}: rank {len(shape)} tensors are not supported")
        names.append(k.encode())
        arrays.append((a, shape))
    n = len(names)
    c_names = (C.c_char_p * max(n, 1))(*names)
    c_data = (C.c_void_p * max(n, 1))(*[a.ctypes.data for a, _ in arrays])
    c_shapes = (C.c_int64 * (4 * max(n, 1)))()
    c_ndims = (C.c_int32 * max(n, 1))()
    c_dtypes = (C.c_int32 * max(n, 1))()
    for i, (a, shape) in enumerate(arrays):
        for d, s in enumerate(shape):
            c_shapes[4 * i + d] = s
        c_ndims[i] = len(shape)
        c_dtypes[i] = _lib.I64 if a.dtype == np.int64 else _lib.F32
    rc = lib.svhip_blob_write(str(path).encode(), model_id(model), n, c_names, c_data, c_shapes, c_ndims, c_dtypes)
    if rc != _lib.OK:
        raise _blob_error(rc)


def read_blob(path):
    """-> (model id, OrderedDict name -> array copy).  Validates magic, version, size and checksum (C reader)."""
    lib = _lib.load()
    b = C.c_void_p()
    rc = lib.svhip_blob_open(str(path).encode(), C.byref(b))
    if rc != _lib.OK:
        raise _blob_error(rc)
    try:
        out = OrderedDict()
        for i in range(lib.svhip_blob_count(b)):
            name, data = C.c_char_p(), C.c_void_p()
            shape = (C.c_int64 * 4)()
            ndim, dtype = C.c_int32(), C.c_int32()
            rc = lib.svhip_blob_tensor(b, i, C.byref(name), C.byref(data), shape, C.byref(ndim), C.byref(dtype))
            if rc != _lib.OK:
                raise _blob_error(rc)
            shp = tuple(shape[d] for d in range(ndim.value))
            np_dt = np.int64 if dtype.value == _lib.I64 else np.float32
            count = int(np.prod(shp, dtype=np.int64)) if shp else 1
            if count:
                buf = (C.c_char * (count * np.dtype(np_dt).itemsize)).from_address(data.value)
                a = np.frombuffer(buf, dtype=np_dt, count=count).reshape(shp).copy()
            else:
                a = np.zeros(shp, np_dt)
            out[name.value.decode()] = a
        return lib.svhip_blob_model(b), out
    finally:
        lib.svhip_blob_close(b)


FUSION_MODELS = {"Raw_ECAPA_sinc_asp": (("ECAPA_TDNN.", "ECAPA_TDNN", ".ecapa"), ("rawnet2v2.", "RawNet2_custom", ".rawnet2"))}


def fusion_blob_paths(dst, model="Raw_ECAPA_sinc_asp"):
    """the two branch blobs a fusion checkpoint converts to: (ECAPA path, RawNet2 path)"""
    return tuple(str(dst) + suffix for _, _, suffix in FUSION_MODELS[model])


def convert_checkpoint(src, dst, model) -> int:
    """Reference checkpoint file (torch pickle) or state dict -> blob at ``dst``.  Returns the number of tensors written.

    A fusion checkpoint (``model='Raw_ECAPA_sinc_asp'``: keys ``__S__.ECAPA_TDNN.*`` / ``__S__.rawnet2v2.*``,
    Raw_ECAPA_sinc_asp.py:22-28) becomes one blob per branch, ``dst + '.ecapa'`` and ``dst + '.rawnet2'`` — a handle is one
    network, and ``Raw_ECAPA.load_blob(dst)`` reads the pair back."""
    if isinstance(src, (str, bytes)) or hasattr(src, "__fspath__"):
        import torch  # host-side only: the one place a pickle is read
        state = torch.load(src, map_location="cpu")     # the reference's 'cpu:0' is rejected by current torch
        if isinstance(state, dict) and "model" in state and "optimizer" in state:   # trainer.py:170-180 full-state files
            state = state["model"]
    else:
        state = src
    sd = embedding_state_dict(state)
    if model in FUSION_MODELS:
        total = 0
        for prefix, branch, suffix in FUSION_MODELS[model]:
            sub = OrderedDict((k[len(prefix):], v) for k, v in sd.items() if k.startswith(prefix))
            if not sub:
                raise ValueError(f"{model}: the checkpoint holds no '{prefix}*' tensors")
            write_blob(str(dst) + suffix, branch, sub)
            total += len(sub)
        return total
    if any(k.startswith(("ECAPA_TDNN.", "rawnet2v2.")) for k in sd):
        raise ValueError("this is a fusion checkpoint (ECAPA_TDNN.* / rawnet2v2.* keys): convert it with "
                         "model='Raw_ECAPA_sinc_asp' (one blob per branch)")
    write_blob(dst, model, sd)
    return len(sd)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--model", default="ECAPA_TDNN", help="reference model name (ECAPA_TDNN, RawNet2_custom, Raw_ECAPA_sinc_asp)")
    a = ap.parse_args(argv)
    n = convert_checkpoint(a.src, a.dst, a.model)
    print(f"{a.dst}: {n} tensors")


if __name__ == "__main__":
    main()
