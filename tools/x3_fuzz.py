"""More seeds of tests/test_gpu_fuzz.py::test_random_geometry_f32x3 (developer tool):  python tools/x3_fuzz.py [n] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from speakerverification_amd import synth
from speakerverification_amd.engine import Engine

n, seed = (int(sys.argv[1]) if len(sys.argv) > 1 else 60), (int(sys.argv[2]) if len(sys.argv) > 2 else 1)
rng = np.random.default_rng(seed)
worst, bad = 0.0, 0
for it in range(n):
    C = int(rng.choice([256, 512, 1024])); T = int(rng.integers(33, 560)); B = int(rng.choice([1, 2, 3, 4, 5, 7, 9, 12]))
    cus = int(rng.choice([0, 1, 2, 3, 5, 7, 11])); sw, sx = int(rng.integers(1, 1000)), int(rng.integers(1, 1000))
    if cus:
        os.environ["SVHIP_PW3_CUS"] = str(cus)
    else:
        os.environ.pop("SVHIP_PW3_CUS", None)
    sd = synth.synth_state_dict(synth.ecapa_param_spec(C=C), seed=sw)
    mel = synth.synth_mel(B, 80, T, seed=sx)
    outs = {}
    for compute in ("f32", "f32x3"):
        eng = Engine(model="ecapa", compute=compute, channels=C, max_batch=B, samples=(T - 1) * 80)
        eng.load_state_dict(sd); eng.finalize()
        outs[compute] = eng.embed_features(mel)
        eng.close()
    scale = float(np.abs(outs["f32"]).max()); err = float(np.abs(outs["f32x3"] - outs["f32"]).max()) / max(1.0, scale)
    worst = max(worst, err)
    ok = np.isfinite(outs["f32x3"]).all() and err <= 1e-4
    bad += (not ok)
    print(f"{it:3d} C={C} T={T} B={B} cus={cus} rel err {err:.2e} {'ok' if ok else 'FAIL'}", flush=True)
print("worst", worst, "failures", bad)
sys.exit(1 if bad else 0)
