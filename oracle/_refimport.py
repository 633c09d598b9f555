"""Import the reference (``/root/reference/src``) in the BUILD CONTAINER ONLY.

Used solely by ``oracle/make_golden.py`` to generate the fixtures under ``tests/golden``.  The
reference never travels to the GPU box; nothing in ``tests/``, ``bench.py`` or the product imports
this module.  Packages the reference imports at module scope but never touches on the hot path
(torchaudio, nnAudio, librosa, soundfile, pydub, onnx, ...) are absent from this image and are
replaced by empty in-memory module objects so that the *reference's own* model / scoring code can
be imported unmodified.  Nothing is written under ``/root/reference``.
"""
from __future__ import annotations

import importlib.machinery
import os
import sys
import types

REF_SRC = "/root/reference/src"

_STUBS = ["torchaudio", "torchsummary", "seaborn", "hyperpyyaml", "onnx", "onnxruntime", "soundfile",
          "pydub", "nnAudio", "nnAudio.features", "nnAudio.features.mel", "librosa", "webrtcvad"]


def reference_available() -> bool:
    return os.path.isdir(REF_SRC)


def import_reference():
    """Returns a namespace with the reference modules on the hot path."""
    if not reference_available():
        raise RuntimeError("reference checkout not present (expected only in the build container)")
    sys.dont_write_bytecode = True
    for name in _STUBS:
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__spec__ = importlib.machinery.ModuleSpec(name, None)
            sys.modules[name] = m
    sys.modules["torchsummary"].summary = lambda *a, **k: None
    sys.modules["hyperpyyaml"].load_hyperpyyaml = lambda *a, **k: {}
    sys.modules["nnAudio"].features = sys.modules["nnAudio.features"]
    sys.modules["nnAudio.features"].mel = sys.modules["nnAudio.features.mel"]
    sys.modules["pydub"].AudioSegment = object
    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)
    ns = types.SimpleNamespace()
    from models import ECAPA_TDNN, RawNet2_custom  # noqa: E402  (reference modules)
    import utils as ref_utils                      # noqa: E402
    ns.ECAPA_TDNN = ECAPA_TDNN
    ns.RawNet2_custom = RawNet2_custom
    ns.utils = ref_utils
    return ns
