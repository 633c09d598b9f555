"""developer experiment: two engines on two streams, alternate full batches (do kernel tails / small kernels of one batch fill under the other's GEMMs?)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from speakerverification_amd.engine import Engine

dev = torch.device("cuda", 0)
B, K = 256, 40
model = sys.argv[1] if len(sys.argv) > 1 else "ecapa"
res = {}
for ns in (1, 2, 3):
    streams = [torch.cuda.Stream(device=dev) for _ in range(ns)]
    engs = []
    for s in streams:
        with torch.cuda.stream(s):
            engs.append(bench.make_engine(model, "bf16", B, 0))
    with torch.cuda.stream(streams[0]):
        wavs = bench.synth_batches(engs[0], 8, B, 0, dev)
    outs = [torch.empty((B, engs[0].embed_dim), device=dev) for _ in range(ns)]
    torch.cuda.synchronize()
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(K):
            i = k % ns
            with torch.cuda.stream(streams[i]):
                engs[i].embed_wave(wavs[k % 8], out=outs[i], async_=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    res[ns] = K * B / dt
    print(model, "streams", ns, "utt/s", round(res[ns]), "ms/step", round(dt / K * 1e3, 4))
    for e in engs: e.close()
