"""developer check: fused 128-channel RawNet2 blocks vs the unfused kernel sequence vs the fp32 engine"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from speakerverification_amd import synth
from speakerverification_amd.engine import Engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 6
sd = synth.synth_state_dict(synth.rawnet2_param_spec(), seed=1)
wav = synth.synth_waveforms(B, 32000, seed=3)
outs = {}
for name, compute, env in (("f32", "f32", None), ("unfused", "bf16", "1"), ("fused", "bf16", None)):
    if env: os.environ["SVHIP_RN_UNFUSED"] = env
    else: os.environ.pop("SVHIP_RN_UNFUSED", None)
    eng = Engine(model="rawnet2", compute=compute, embed_dim=320, max_batch=B)
    eng.load_state_dict(sd); eng.finalize()
    outs[name] = eng.embed_wave(wav).reshape(B, -1)
    eng.embed_wave(wav)
    t0 = time.perf_counter()
    for _ in range(3): eng.embed_wave(wav)
    print(name, "ms/call", (time.perf_counter() - t0) / 3 * 1e3)
    eng.close()
ref = outs["f32"]
for k in ("unfused", "fused"):
    a = outs[k]
    cos = (a * ref).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(ref, axis=1))
    print(k, "finite", np.isfinite(a).all(), "cos min", cos.min(), "rel", np.abs(a - ref).max() / np.abs(ref).max())
a, b = outs["fused"], outs["unfused"]
print("fused vs unfused: rel", np.abs(a - b).max() / np.abs(b).max())
