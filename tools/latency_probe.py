"""developer probe: per-call time of small batches (the reference API embeds B = num_eval = 10 - 20 crops of one file per call), option A/B:
   python tools/latency_probe.py [model=ecapa|rawnet2] [compute=bf16] [opt=val ...]"""
import sys, time
import torch
sys.path.insert(0, ".")
import bench
kv = dict(a.split("=") for a in sys.argv[1:])
model, compute = kv.pop("model", "ecapa"), kv.pop("compute", "bf16")
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev)
with torch.cuda.stream(st):
    eng = bench.make_engine(model, compute, 256, 0)
    wavs = bench.synth_batches(eng, 1, 256, 0, dev)
    for rnd in range(3):
        for on in (0, 1):
            for k, v in kv.items():
                eng.set_option(k, int(v) if on else 0)
            row = []
            for b in (10, 20, 32, 64):
                w = wavs[0][:b].contiguous()
                out = torch.empty((b, eng.embed_dim), device=dev)
                for _ in range(5):
                    eng.embed_wave(w, out=out, async_=True)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(50):
                    eng.embed_wave(w, out=out, async_=True)
                torch.cuda.synchronize()
                row.append(f"B={b}: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms")
            print("opts", "on " if on else "off", "  ".join(row), flush=True)
    if "prof" in sys.argv[0] or True:
        w = wavs[0][:20].contiguous(); out = torch.empty((20, eng.embed_dim), device=dev)
        for k, v in kv.items():
            eng.set_option(k, 0)
        eng.profile(True); eng.embed_wave(w, out=out, async_=True); torch.cuda.synchronize()
        print({k: round(v["ms"] * 1e3, 1) for k, v in sorted(eng.profile_results().items(), key=lambda kv: -kv[1]["ms"])})
