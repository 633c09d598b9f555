"""fbank kernel alone at B = 256 (bf16-handle split form and exact fp32 form): median of 30 launches, HIP events on the engine's stream."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speakerverification_amd.engine import Engine
dev = torch.device("cuda", 0)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for compute in ("bf16", "f32"):
        eng = Engine(model="ecapa", compute=compute, channels=64, max_batch=256, stream=torch.cuda.current_stream().cuda_stream)
        wav = torch.randn((256, 32000), device=dev) * 0.1
        mel = torch.empty((256, 80, 401), device=dev)
        for _ in range(5):
            eng.fbank(wav, out=mel, async_=True)
        ts = []
        for _ in range(30):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); eng.fbank(wav, out=mel, async_=True); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        print(f"fbank {compute}: median {statistics.median(ts):.1f} us, min {min(ts):.1f} us")
        eng.close()
