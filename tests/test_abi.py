"""CPU: the C-ABI library builds, loads and exports every symbol include/svhip.h declares; the
product path fails loudly (no CPU fallback) when no GPU / no library is there."""
import ctypes
import os
import re

import pytest

from speakerverification_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "svhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(svhip_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert header_symbols() == _lib.exported_symbols()


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    for name in header_symbols():
        assert hasattr(lib, name), name
    assert lib.svhip_abi_version() == 5


def test_default_config_matches_reference_defaults():
    cfg = _lib.default_config()
    assert cfg.struct_size == ctypes.sizeof(_lib.Config)
    # feature.py:66-71 defaults, ECAPA_TDNN.py:378 channels, 2 s @ 16 kHz
    assert (cfg.fb_sr, cfg.n_fft, cfg.win_length, cfg.hop_length, cfg.n_mels) == (8000, 512, 200, 80, 80)
    assert (cfg.channels, cfg.embed_dim, cfg.samples) == (1024, 192, 32000)
    assert abs(cfg.preemph - 0.97) < 1e-7 and cfg.fmin == 0.0 and cfg.fmax < 0


def test_create_rejects_bad_config_without_touching_a_gpu():
    lib = _lib.load()
    cfg = _lib.default_config()
    cfg.struct_size = 4
    h = ctypes.c_void_p()
    assert lib.svhip_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    assert b"struct_size" in lib.svhip_last_error(None)


def test_no_cpu_fallback():
    """Without a GPU the product raises; it never routes through the oracle."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from speakerverification_amd.engine import Engine
    with pytest.raises(_lib.SvhipError):
        Engine(model="none")
    import speakerverification_amd
    pkg = os.path.dirname(speakerverification_amd.__file__)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_per_device_launch_attribute_bookkeeping():
    """hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device; the launchers keep one flag bit per device
    ordinal (csrc/kernels.h DeviceOnce).  svhip_selftest exercises that bookkeeping on the host."""
    assert _lib.load().svhip_selftest() == 0
    src = os.path.join(ROOT, "speakerverification_amd", "csrc")
    for f in os.listdir(src):
        if f.endswith(".hip"):
            text = open(os.path.join(src, f)).read()
            assert "static bool attr" not in text, f        # no process-wide flags left
            if "hipFuncSetAttribute" in text:
                raise AssertionError(f"{f}: raise LDS limits through set_max_dynamic_lds (per-device)")


def test_comm_entry_points_fail_cleanly_without_a_handle():
    lib = _lib.load()
    assert lib.svhip_comm_init(None, None, 0, 1) == -1
    assert lib.svhip_allgather_rows(None, None, 0, 0, None, 0) == -1
    assert lib.svhip_comm_destroy(None) == -1


def test_comm_reports_a_missing_rccl_instead_of_crashing():
    """with no loadable RCCL the comm entry points must return SVHIP_ERR_UNSUPPORTED and leave a message
    (svhip.h's promise), not crash while building that message.  The loader runs once per process, so: a child process with
    SVHIP_RCCL_LIB pointing at nothing."""
    import subprocess
    import sys
    code = (
        "import ctypes as C, sys\n"
        "sys.path.insert(0, %r)\n"
        "from speakerverification_amd import _lib\n"
        "lib = _lib.load()\n"
        "buf = C.create_string_buffer(128)\n"
        "rc = lib.svhip_comm_unique_id(buf)\n"
        "msg = lib.svhip_comm_last_error()\n"
        "print(rc, (msg or b'').decode())\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SVHIP_RCCL_LIB="/nonexistent/librccl.so.1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    rc, _, msg = r.stdout.strip().partition(" ")
    assert int(rc) == -5, r.stdout                      # SVHIP_ERR_UNSUPPORTED
    assert "cannot load librccl" in msg and "nonexistent" in msg
