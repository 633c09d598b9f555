"""Developer tool: random (model, channels, utterance length, batch) geometries — the bf16 engine against the fp32 engine.
Every dispatch decision (kernel routes by grid size, fused / separate RawNet2 tails, conv-gather with and without the appended
shortcut segment, Res2Net chain vs per-layer GEMMs) depends on these numbers.   python tools/geometry_fuzz.py [n] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from speakerverification_amd import synth
from speakerverification_amd.engine import Engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(n):
    model = "rawnet2" if rng.random() < 0.5 else "ecapa"
    B = int(rng.choice([1, 2, 3, 5, 8, 17, 33, 64]))
    if model == "rawnet2":
        L = int(rng.integers(6000, 52000))
        kw, spec = dict(embed_dim=320), synth.rawnet2_param_spec(nb_samp=L)
        bar = 0.985
    else:
        C = int(rng.choice([64, 128, 192, 256, 512, 1024]))
        L = int(rng.integers(50, 520)) * 80              # the fbank frames an utterance every 80 samples
        kw, spec = dict(channels=C), synth.ecapa_param_spec(C=C)
        bar = 0.998
    sd = synth.synth_state_dict(spec, seed=int(rng.integers(1, 1000)))
    wav = synth.synth_waveforms(B, L, seed=int(rng.integers(1, 1000)))
    outs = {}
    # RawNet2 on random weights: the bf16 engine can sit a few percent from the fp32 one (8 un-normalised residual blocks), with
    # the separate kernel sequence exactly as much as with the fused kernels — so the fused path is held against the separate one
    runs = [("f32", "f32", False), ("bf16", "bf16", False)] + ([("sep", "bf16", True)] if model == "rawnet2" else [])
    for name, compute, separate in runs:
        if separate:
            os.environ["SVHIP_RN_UNFUSED"] = "1"
        else:
            os.environ.pop("SVHIP_RN_UNFUSED", None)
        eng = Engine(model=model, compute=compute, max_batch=B, samples=L, **kw)
        eng.load_state_dict(sd)
        eng.finalize()
        outs[name] = eng.embed_wave(wav).reshape(B, -1)
        eng.close()
    os.environ.pop("SVHIP_RN_UNFUSED", None)

    def cosine(a, b):
        return (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))
    c32 = cosine(outs["f32"], outs["bf16"])
    ok = bool(np.isfinite(outs["bf16"]).all())
    if model == "rawnet2":
        csep = cosine(outs["sep"], outs["bf16"])
        ok = ok and csep.min() >= 0.999 and c32.min() >= 0.9
        extra = f" vs separate kernels {csep.min():.5f}"
    else:
        ok = ok and c32.min() >= bar
        extra = ""
    bad += not ok
    print(f"{'ok ' if ok else 'BAD'} {model} {kw} L={L} B={B} min cos vs fp32 {c32.min():.5f}{extra}", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
