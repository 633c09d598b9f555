import os, sys
sys.path.insert(0, '.')
import numpy as np, torch
from speakerverification_amd import synth
from speakerverification_amd.engine import Engine
from oracle import rawnet2 as o_rn, ecapa as o_e
sd = synth.synth_state_dict(synth.rawnet2_param_spec(nOut=320), seed=1)
x = synth.synth_waveforms(2, 32000, seed=20220829)
st = {}
with torch.no_grad():
    ref = o_rn.rawnet2_forward(torch.from_numpy(x), o_e.to_torch_sd(sd), stages=st)
names = ["front", "layer1", "layer2", "layer3", None, "layer4", "layer5", None, "layer6"]
for stop in (0, 1, 2, 3, 5, 6, 8):
    os.environ["SVHIP_RN_STOP"] = str(stop)
    eng = Engine(model="rawnet2", compute="f32", embed_dim=320, max_batch=2, samples=32000)
    eng.load_state_dict(sd); eng.finalize()
    eng.embed_wave(x)
    a = eng.get_stage("rn_x")
    r = st[names[stop]].numpy()
    B, C, T = r.shape
    got = a.reshape(B, T, C).transpose(0, 2, 1)
    print(names[stop], r.shape, "max|d|", np.abs(got - r).max(), "scale", np.abs(r).max())
    eng.close()
# filters
f = st["sinc_filters"].numpy(); print("filt range", f.min(), f.max())
