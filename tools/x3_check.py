"""developer check: the F32X3 compute mode (split-bf16 GEMMs on fp32 operands) against the golden fp32 fixtures + speed"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from speakerverification_amd import synth
from speakerverification_amd.engine import Engine

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
for C in (512, 1024):
    g = np.load(os.path.join(G, f"ecapa_C{C}_T401.npz"))
    B, T = int(g["B"]), int(g["T"])
    mel = synth.synth_mel(B, 80, T, seed=int(g["seed_x"]))
    sd = synth.synth_state_dict(synth.ecapa_param_spec(C=C), seed=int(g["seed_w"]))
    ref = g["out"]
    for compute in ("f32", "f32x3"):
        eng = Engine(model="ecapa", compute=compute, channels=C, max_batch=B)
        eng.load_state_dict(sd); eng.finalize()
        out = eng.embed_features(mel)
        nrm = lambda a: a / np.maximum(np.linalg.norm(a, axis=1, keepdims=True), 1e-12)
        print(f"C={C} {compute}: max|d| {np.abs(out-ref).max():.3e} (scale {np.abs(ref).max():.1f}, rel {np.abs(out-ref).max()/np.abs(ref).max():.2e}); normalised abs {np.abs(nrm(out)-nrm(ref)).max():.3e}")
        eng.close()
wav = synth.synth_waveforms(256)
for compute in ("f32", "f32x3"):
    eng = Engine(model="ecapa", compute=compute, channels=1024, max_batch=256)
    eng.load_state_dict(synth.synth_state_dict(synth.ecapa_param_spec(C=1024), seed=1)); eng.finalize()
    eng.embed_wave(wav)
    t0 = time.perf_counter()
    for _ in range(3): eng.embed_wave(wav)
    dt = (time.perf_counter() - t0) / 3
    print(compute, "B=256 ms/step", dt * 1e3, "utt/s", 256 / dt)
    eng.close()
