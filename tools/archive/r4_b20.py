"""developer probe (round 4): ECAPA bf16 B = 20 call time against the Res2Net slice count (option r2_slices)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
st = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(st)
eng = bench.make_engine("ecapa", "bf16", 256, 0)
wavs = bench.synth_batches(eng, 2, 256, 0, dev)
B = 20
w = wavs[0][:B].contiguous()
out = torch.empty((B, eng.embed_dim), device=dev, dtype=torch.float32)
for S in (-1, 3, 4, 5, 6, 7, 8, -1):
    try:
        eng.set_option("r2_slices", S)
        for _ in range(5): eng.embed_wave(w, out=out, async_=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); n = 200
        for _ in range(n): eng.embed_wave(w, out=out, async_=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        eng.profile(True); eng.embed_wave(w, out=out, async_=True); torch.cuda.synchronize()
        pr = eng.profile_results(); eng.profile(False)
        r2 = sum(v["ms"] for k, v in pr.items() if "res2" in k)
        print("r2_slices", S, ": %.3f ms per call, res2net %.1f us" % (dt * 1e3, r2 * 1e3), flush=True)
    except Exception as e:
        print("r2_slices", S, "failed:", e, flush=True)
