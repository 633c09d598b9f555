"""developer probe: RawNet2 fp16 at B = 256 with the symmetric sinc form (default) and the 251-tap kernel (option rn_sinc_full): per-kernel time,
embeddings against each other and against the exact-fp32 handle on 8 rows"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from speakerverification_amd import synth
from speakerverification_amd.engine import Engine
dev = torch.device("cuda", 0)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    sd = synth.synth_state_dict(synth.rawnet2_param_spec(nOut=320), seed=1)
    eng = Engine(model="rawnet2", compute="f16", embed_dim=320, max_batch=256, samples=32000, stream=torch.cuda.current_stream().cuda_stream)
    eng.load_state_dict(sd); eng.finalize()
    wav = torch.from_numpy(synth.synth_waveforms(256, 32000, seed=5)).cuda()
    out = {}
    for rnd in range(2):
        for mode in (0, 1):
            eng.set_option("rn_sinc_full", mode)
            o = torch.empty((256, 320), device=dev)
            for _ in range(3): eng.embed_wave(wav, out=o, async_=True)
            torch.cuda.synchronize()
            eng.profile(True)
            for _ in range(10): eng.embed_wave(wav, out=o, async_=True)
            torch.cuda.synchronize()
            p = eng.profile_results(); eng.profile(False)
            tot = sum(v["ms"] for v in p.values()) / 10
            print(f"rn_sinc_full {mode}: rn_sinc {p['rn_sinc']['ms'] / p['rn_sinc']['launches'] * 1e3:7.1f} us; all kernels {tot:.3f} ms per step", flush=True)
            out[mode] = o.cpu().numpy()
    ref = Engine(model="rawnet2", compute="f32", embed_dim=320, max_batch=8, samples=32000)
    ref.load_state_dict(sd); ref.finalize()
    r = ref.embed_wave(wav[:8].cpu().numpy())
    for mode in (0, 1):
        a = out[mode][:8]
        cos = np.sum(a * r, axis=1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(r, axis=1))
        print(f"rn_sinc_full {mode}: vs exact f32: cos >= {cos.min():.6f}, max err / scale {np.abs(a - r).max() / np.abs(r).max():.2e}")
    a, b = out[0], out[1]
    cos = np.sum(a * b, axis=1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))
    print(f"symmetric vs 251-tap: cos >= {cos.min():.6f}, max diff / scale {np.abs(a - b).max() / np.abs(b).max():.2e}")
