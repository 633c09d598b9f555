import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, numpy as np, torch
from speakerverification_amd import synth
from speakerverification_amd.engine import Engine
for C in (512,):
    for B, L in ((256, 32000), (64, 48000), (256, 16000)):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            eng = Engine(model="ecapa", compute="bf16", channels=C, max_batch=B, samples=L, stream=st.cuda_stream)
            eng.load_state_dict(synth.synth_state_dict(synth.ecapa_param_spec(C=C), seed=1)); eng.finalize()
            wav = torch.from_numpy(synth.synth_waveforms(B, L, seed=1)).cuda()
            out = torch.empty((B, 192), device="cuda")
            for _ in range(3): eng.embed_wave(wav, out=out, async_=True)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): eng.embed_wave(wav, out=out, async_=True)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
            eng.profile(True)
            eng.embed_wave(wav, out=out, async_=True); torch.cuda.synchronize()
            pr = eng.profile_results(); eng.profile(False)
            print(f"C={C} B={B} L={L}: {B/dt:.0f} utt/s ({dt*1e3:.2f} ms)", {k: round(v['ms'],3) for k,v in pr.items()})
            eng.close()
