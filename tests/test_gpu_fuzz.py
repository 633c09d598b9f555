"""GPU: seeded random geometries (model, channels, utterance length, batch).  Every dispatch decision of the library — kernel route
by grid size, fused / separate RawNet2 block tails, conv-gather with and without the appended shortcut segment, Res2Net chain vs
per-layer GEMMs, persistent sinc kernel item ranges — depends on these numbers; tools/geometry_fuzz.py runs more of them."""
import numpy as np
import pytest

from speakerverification_amd import synth
from speakerverification_amd.engine import Engine

pytestmark = pytest.mark.gpu


def _cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        model = "rawnet2" if rng.random() < 0.5 else "ecapa"
        B = int(rng.choice([1, 2, 3, 5, 8, 17, 33]))
        if model == "rawnet2":
            out.append((model, 0, int(rng.integers(6000, 52000)), B, int(rng.integers(1, 1000)), int(rng.integers(1, 1000))))
        else:
            out.append((model, int(rng.choice([64, 128, 192, 256, 512, 1024])), int(rng.integers(50, 520)) * 80, B,
                        int(rng.integers(1, 1000)), int(rng.integers(1, 1000))))
    return out


@pytest.mark.parametrize("model,C,L,B,sw,sx", _cases(10, 2024))
def test_random_geometry(model, C, L, B, sw, sx, monkeypatch):
    if model == "rawnet2":
        kw, spec = dict(embed_dim=320), synth.rawnet2_param_spec(nb_samp=L)
        runs = [("f32", "f32", False), ("bf16", "bf16", False), ("sep", "bf16", True)]
    else:
        kw, spec = dict(channels=C), synth.ecapa_param_spec(C=C)
        runs = [("f32", "f32", False), ("bf16", "bf16", False)]
    sd = synth.synth_state_dict(spec, seed=sw)
    wav = synth.synth_waveforms(B, L, seed=sx)
    outs = {}
    for name, compute, separate in runs:
        if separate:
            monkeypatch.setenv("SVHIP_RN_UNFUSED", "1")
        else:
            monkeypatch.delenv("SVHIP_RN_UNFUSED", raising=False)
        eng = Engine(model=model, compute=compute, max_batch=B, samples=L, **kw)
        eng.load_state_dict(sd)
        eng.finalize()
        outs[name] = eng.embed_wave(wav).reshape(B, -1)
        eng.close()

    def cosine(a, b):
        return (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))
    assert np.isfinite(outs["bf16"]).all()
    c32 = cosine(outs["f32"], outs["bf16"])
    if model == "rawnet2":
        # on random weights the bf16 engine sits a few percent from the fp32 one, with the separate kernel sequence exactly as much
        # as with the fused kernels (8 un-normalised residual blocks): the fused path is held against the separate one
        assert cosine(outs["sep"], outs["bf16"]).min() >= 0.999 and c32.min() >= 0.9, c32
    else:
        assert c32.min() >= 0.998, c32
