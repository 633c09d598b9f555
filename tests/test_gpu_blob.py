"""GPU: a handle loaded from a packed checkpoint blob (svhip_load_blob: mmap -> load_tensor -> finalize inside the
library) is bit-identical to one loaded tensor by tensor from the state dict; wrong-model and damaged blobs fail loudly."""
import numpy as np
import pytest

from speakerverification_amd import _lib, checkpoint, synth
from speakerverification_amd.engine import Engine

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("model,compute", [("ecapa", "f32"), ("ecapa", "bf16"), ("rawnet2", "f32")])
def test_blob_loaded_handle_is_bit_identical(tmp_path, model, compute):
    if model == "ecapa":
        spec, kw = synth.ecapa_param_spec(C=512), dict(channels=512)
    else:
        spec, kw = synth.rawnet2_param_spec(), dict(embed_dim=320)
    sd = synth.synth_state_dict(spec, seed=3)
    p = tmp_path / "w.svhip"
    checkpoint.write_blob(p, model, sd)
    wav = synth.synth_waveforms(4, 32000, seed=11)
    a = Engine(model=model, compute=compute, max_batch=4, **kw)
    a.load_state_dict(sd)
    a.finalize()
    b = Engine(model=model, compute=compute, max_batch=4, **kw)
    b.load_blob(p)
    ea, eb = a.embed_wave(wav), b.embed_wave(wav)
    assert np.array_equal(ea, eb)
    a.close(); b.close()


def test_blob_errors_reach_the_handle(tmp_path):
    sd = synth.synth_state_dict(synth.ecapa_param_spec(C=64), seed=3)
    p = tmp_path / "w.svhip"
    checkpoint.write_blob(p, "rawnet2", sd)                     # wrong model id
    e = Engine(model="ecapa", channels=64, max_batch=2)
    with pytest.raises(_lib.SvhipError, match="model"):
        e.load_blob(p)
    checkpoint.write_blob(p, "ecapa", {k: v for k, v in sd.items() if k != "fc.conv.bias"})
    with pytest.raises(_lib.SvhipError, match="fc.conv.bias"):   # a missing tensor is named
        e.load_blob(p)
    e.close()
    e = Engine(model="ecapa", channels=64, max_batch=2)
    raw = bytearray(p.read_bytes()); raw[-5] ^= 1; p.write_bytes(bytes(raw))
    with pytest.raises(_lib.SvhipError, match="checksum"):
        e.load_blob(p)
    e.close()


def test_model_handling_loads_a_blob(tmp_path):
    """ModelHandling.loadParameters (model.py:718-746 counterpart) accepts the blob in place of the torch pickle."""
    import torch
    from speakerverification_amd.model import ModelHandling, SpeakerEncoder, WrappedModel
    from tests.test_gpu_e2e import ARGS as args
    sd = synth.synth_state_dict(synth.ecapa_param_spec(C=512), seed=1)
    blob, pt = tmp_path / "w.svhip", tmp_path / "w.model"
    checkpoint.write_blob(blob, "ECAPA_TDNN", sd)
    torch.save({"__S__." + k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, pt)
    wav = synth.synth_waveforms(2, 32000, seed=4)
    outs = []
    for path in (blob, pt):
        mh = ModelHandling(WrappedModel(SpeakerEncoder(**args)), **args)
        mh.loadParameters(str(path), show_error=False)
        outs.append(mh.embed_utterance(wav[0], num_eval=2, normalize=True).numpy())
    assert np.array_equal(outs[0], outs[1])
