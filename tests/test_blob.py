"""CPU: the packed checkpoint blob (csrc/blob.hip) — C writer / reader round trip, rejection of damaged files, and the
host-side converter from a reference-layout checkpoint (module.__S__.* keys; trainer.py:145-205, model.py:718-746)."""
import os

import numpy as np
import pytest
import torch

from speakerverification_amd import _lib, checkpoint, synth


def small_state():
    return synth.synth_state_dict(synth.ecapa_param_spec(C=64), seed=5)


def test_roundtrip_is_bit_exact(tmp_path):
    sd = small_state()
    p = tmp_path / "w.svhip"
    checkpoint.write_blob(p, "ECAPA_TDNN", sd)
    assert checkpoint.is_blob(p)
    mid, back = checkpoint.read_blob(p)
    assert mid == _lib.MODEL_ECAPA
    assert list(back) == list(sd)                                  # order kept
    for k, v in sd.items():
        assert back[k].shape == np.shape(v), k                      # 0-d num_batches_tracked stays 0-d
        assert back[k].dtype == (np.int64 if np.asarray(v).dtype == np.int64 else np.float32), k
        assert np.array_equal(back[k], np.asarray(v)), k


def test_empty_and_zero_sized(tmp_path):
    p = tmp_path / "e.svhip"
    checkpoint.write_blob(p, "rawnet2", {})
    mid, back = checkpoint.read_blob(p)
    assert mid == _lib.MODEL_RAWNET2 and len(back) == 0
    checkpoint.write_blob(p, "ecapa", {"a": np.zeros((0, 3), np.float32), "b": np.arange(6, dtype=np.float32).reshape(1, 2, 3, 1)})
    _, back = checkpoint.read_blob(p)
    assert back["a"].shape == (0, 3) and back["b"].shape == (1, 2, 3, 1) and back["b"].ravel().tolist() == list(range(6))


@pytest.mark.parametrize("damage", ["flip_payload", "truncate", "magic", "version", "append"])
def test_damaged_files_are_rejected(tmp_path, damage):
    p = tmp_path / "w.svhip"
    checkpoint.write_blob(p, "ECAPA_TDNN", small_state())
    raw = bytearray(p.read_bytes())
    if damage == "flip_payload":
        raw[len(raw) // 2] ^= 0x40
    elif damage == "truncate":
        raw = raw[:-100]
    elif damage == "magic":
        raw[0] = ord("X")
    elif damage == "version":
        raw[8] = 9
    elif damage == "append":
        raw += b"\0" * 64
    p.write_bytes(bytes(raw))
    with pytest.raises(_lib.SvhipError) as ei:
        checkpoint.read_blob(p)
    msg = str(ei.value)
    want = {"flip_payload": "checksum", "truncate": "truncated", "magic": "magic", "version": "version", "append": "truncated or padded"}[damage]
    assert want in msg, msg


def test_missing_file_and_bad_rank(tmp_path):
    with pytest.raises(_lib.SvhipError):
        checkpoint.read_blob(tmp_path / "nope.svhip")
    with pytest.raises(ValueError):
        checkpoint.write_blob(tmp_path / "r.svhip", "ecapa", {"x": np.zeros((1, 1, 1, 1, 1), np.float32)})
    with pytest.raises(ValueError):
        checkpoint.model_id("resnet")


def test_convert_reference_layout_checkpoint(tmp_path):
    """A '.model' file as trainer.py writes it: the WrappedModel state dict with module.__S__ / __L__ / front-end keys."""
    sd = small_state()
    full = {"module.__S__." + k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    full["module.__L__.W"] = torch.zeros(4, 3)
    full["module.compute_features.0.flipped_filter"] = torch.zeros(1, 1, 2)
    src, dst = tmp_path / "model_000000001.model", tmp_path / "w.svhip"
    torch.save(full, src)
    n = checkpoint.convert_checkpoint(str(src), dst, "ECAPA_TDNN")
    assert n == len(sd)
    _, back = checkpoint.read_blob(dst)
    assert set(back) == set(sd)
    for k in sd:
        assert np.array_equal(back[k], np.asarray(sd[k])), k
    # command line entry point
    checkpoint.main([str(src), str(tmp_path / "w2.svhip"), "--model", "ECAPA_TDNN"])
    assert (tmp_path / "w2.svhip").read_bytes() == dst.read_bytes()


def test_model_handling_reads_blobs_on_the_host(tmp_path):
    """HipModule.load_blob fills the host-side state dict (no GPU touched until the first forward)."""
    from speakerverification_amd.models import ECAPA_TDNN
    sd = small_state()
    p = tmp_path / "w.svhip"
    checkpoint.write_blob(p, "ECAPA_TDNN", sd)
    m = ECAPA_TDNN.MainModel(nOut=192, channels=[64, 64, 64, 64, 192], features="melspectrogram")
    m.load_blob(p)
    got = m.state_dict()
    for k, v in sd.items():
        assert np.array_equal(np.asarray(got[k]), np.asarray(v)), k
    with pytest.raises(ValueError):
        checkpoint.write_blob(p, "rawnet2", sd)
        m.load_blob(p)
