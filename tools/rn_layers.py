"""Developer tool: per-GEMM-shape profile rows of one step (SVHIP_LAYER_LABELS=1).   python tools/rn_layers.py [rawnet2|ecapa] [compute] [opt=v,...] [B]"""
import os, sys
os.environ["SVHIP_LAYER_LABELS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from speakerverification_amd import synth
from speakerverification_amd.engine import Engine

model = sys.argv[1] if len(sys.argv) > 1 else "rawnet2"
compute = sys.argv[2] if len(sys.argv) > 2 else ("f16" if model == "rawnet2" else "bf16")
B = int(sys.argv[4]) if len(sys.argv) > 4 else 256
kw = dict(embed_dim=320) if model == "rawnet2" else dict(channels=1024)
eng = Engine(model=model, compute=compute, max_batch=B, **kw)
if len(sys.argv) > 3:
    for kv in sys.argv[3].split(","):          # e.g. cv_off=1,pw3_cus=64
        if kv:
            k, v = kv.split("=")
            eng.set_option(k, int(v))
spec = synth.rawnet2_param_spec(nOut=320) if model == "rawnet2" else synth.ecapa_param_spec(C=1024)
eng.load_state_dict(synth.synth_state_dict(spec, seed=1))
eng.finalize()
wav = torch.from_numpy(synth.synth_waveforms(B, 32000, seed=3)).cuda()
for _ in range(3):
    eng.embed_wave(wav)
eng.profile(True)
N = 5
for _ in range(N):
    eng.embed_wave(wav)
torch.cuda.synchronize()
rows = eng.profile_results()
eng.profile(False)
tot = 0.0
for name, r in rows.items():
    ms = r["ms"] / N
    tot += ms
    print(f"{name:44s} {r['launches'] // N:3d} x  {ms * 1e3:8.1f} us  {r['flops'] / N / (ms * 1e-3) / 1e12 if r['flops'] else 0:7.1f} TF")
print("total %.3f ms" % tot)
