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


def test_fusion_checkpoint_converts_to_one_blob_per_branch(tmp_path):
    """the production model (Raw_ECAPA_sinc_asp, keys __S__.ECAPA_TDNN.* / __S__.rawnet2v2.*) has a blob path:
    one blob per branch; a fusion state dict handed to a single-network conversion is refused with a clear message."""
    e = synth.synth_state_dict(synth.ecapa_param_spec(C=64), seed=5)
    r = synth.synth_state_dict(synth.rawnet2_param_spec(nOut=320), seed=6)
    state = {"module.__S__.ECAPA_TDNN." + k: torch.from_numpy(np.asarray(v)) for k, v in e.items()}
    state.update({"module.__S__.rawnet2v2." + k: torch.from_numpy(np.asarray(v)) for k, v in r.items()})
    state["module.__L__.weight"] = torch.zeros(3)
    state["module.compute_features.0.flipped_filter"] = torch.zeros(1, 1, 2)
    dst = tmp_path / "fusion.svhip"
    n = checkpoint.convert_checkpoint(state, dst, "Raw_ECAPA_sinc_asp")
    assert n == len(e) + len(r)
    p_e, p_r = checkpoint.fusion_blob_paths(dst)
    mid_e, back_e = checkpoint.read_blob(p_e)
    mid_r, back_r = checkpoint.read_blob(p_r)
    assert (mid_e, mid_r) == (_lib.MODEL_ECAPA, _lib.MODEL_RAWNET2)
    assert list(back_e) == list(e) and list(back_r) == list(r)
    assert all(np.array_equal(back_r[k], np.asarray(v)) for k, v in r.items())
    with pytest.raises(ValueError, match="fusion checkpoint"):
        checkpoint.convert_checkpoint(state, tmp_path / "x.svhip", "ECAPA_TDNN")
    with pytest.raises(ValueError, match="holds no"):
        checkpoint.convert_checkpoint({"module.__S__." + k: v for k, v in e.items()}, tmp_path / "y.svhip", "Raw_ECAPA_sinc_asp")


def test_crafted_table_cannot_wrap_the_bounds_check(tmp_path):
    """u64 + u64 / shape products in the table checks must not wrap.  The checksum is no integrity guarantee, so the
    test re-computes it (csrc/blob.hip: four-lane FNV-1a over everything after the 64-byte header) after corrupting an entry."""
    M = 0xFFFFFFFFFFFFFFFF
    prime = 0x100000001b3

    def checksum(b):
        h = [0xcbf29ce484222325, 0x84222325cbf29ce4, 0x9ce484222325cbf2, 0x2325cbf29ce48422]
        i = 0
        while i + 32 <= len(b):
            for lane in range(4):
                w = int.from_bytes(b[i + 8 * lane:i + 8 * lane + 8], "little")
                h[lane] = ((h[lane] ^ w) * prime) & M
            i += 32
        r = 0xcbf29ce484222325
        for lane in range(4):
            r = ((r ^ h[lane]) * prime) & M
        for x in b[i:]:
            r = ((r ^ x) * prime) & M
        return r

    p = tmp_path / "w.svhip"
    checkpoint.write_blob(p, "ECAPA_TDNN", {"a": np.arange(8, dtype=np.float32)})
    raw = bytearray(p.read_bytes())
    assert int.from_bytes(raw[32:40], "little") == checksum(raw[64:])        # header: checksum at byte 32, payload from byte 64
    ent = 64                                                                  # entry 0: data_off at +48, nbytes at +56, shape[0] at +16
    assert int.from_bytes(raw[ent + 56:ent + 64], "little") == 32 and int.from_bytes(raw[ent + 16:ent + 24], "little") == 8
    cases = {
        "nbytes wraps data_off + nbytes": [(ent + 56, 2 ** 64 - 64)],
        "shape product wraps to the stored size": [(ent + 16, 2 ** 62 + 8)],              # (2^62 + 8) * 4 bytes == 32 mod 2^64
        "data_off beyond the file": [(ent + 48, 2 ** 63)],
    }
    for why, edits in cases.items():
        bad = bytearray(raw)
        for off, val in edits:
            bad[off:off + 8] = val.to_bytes(8, "little")
        bad[32:40] = checksum(bad[64:]).to_bytes(8, "little")
        p.write_bytes(bytes(bad))
        with pytest.raises(_lib.SvhipError) as ei:
            checkpoint.read_blob(p)
        assert "checksum" not in str(ei.value), (why, str(ei.value))          # rejected by the table checks, not by luck
