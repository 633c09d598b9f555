"""developer probe: the fused front-end of a bf16 handle at B = 256 with phases ablated (option ff_abl), HIP-event time of the label"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speakerverification_amd import synth
from speakerverification_amd.engine import Engine
dev = torch.device("cuda", 0)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    eng = Engine(model="ecapa", compute="bf16", channels=64, max_batch=256, stream=torch.cuda.current_stream().cuda_stream, on_numeric="ignore")
    eng.load_state_dict(synth.synth_state_dict(synth.ecapa_param_spec(C=64), seed=1)); eng.finalize()
    wav = torch.randn((256, 32000), device=dev) * 0.1
    out = torch.empty((256, 192), device=dev)
    names = {0: "all", 1: "no sample loads", 2: "no operand build", 4: "no MFMAs", 8: "no mel/log", 16: "no row stores", 32: "no norm launch",
             3: "no loads, no build", 12: "no MFMA, no mel", 31: "barriers + staging only", 63: "everything off"}
    for rnd in range(2):
        for abl, nm in names.items():
            eng.set_option("ff_abl", abl)
            for _ in range(3): eng.embed_wave(wav, out=out, async_=True)
            torch.cuda.synchronize()
            eng.profile(True)
            for _ in range(10): eng.embed_wave(wav, out=out, async_=True)
            torch.cuda.synchronize()
            p = eng.profile_results(); eng.profile(False)
            print(f"ff_abl {abl:2d} ({nm:26s}): fbank_fused {p['fbank_fused']['ms'] / p['fbank_fused']['launches'] * 1e3:7.1f} us", flush=True)
