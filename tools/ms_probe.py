"""developer probe (round 4): ECAPA bf16, one engine at B = 256 vs two engines on two streams (alternating whole batches, and the
halves of every batch).  python tools/ms_probe.py"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda:0")
B = 256
main_stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(main_stream)
eng = bench.make_engine("ecapa", "bf16", B, 0)
wavs = bench.synth_batches(eng, 8, B, 0, dev)
out = torch.empty((B, eng.embed_dim), device=dev, dtype=torch.float32)

def run(fn, steps=40):
    for _ in range(5): fn(0)
    torch.cuda.synchronize()
    best = None
    for rep in range(3):
        t0 = time.perf_counter()
        for k in range(steps): fn(k)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        best = dt if best is None or dt < best else best
    return best

t1 = run(lambda k: eng.embed_wave(wavs[k % 8], out=out, async_=True))
print("one engine B=256: %.3f ms/step %.0f utt/s" % (t1 * 1e3, B / t1), flush=True)
for nst, b in ((2, 256), (2, 128), (3, 256), (4, 64)):
    streams = [torch.cuda.Stream(device=dev) for _ in range(nst)]
    engs, outs = [], []
    for st in streams:
        with torch.cuda.stream(st):
            engs.append(bench.make_engine("ecapa", "bf16", b, 0))
            outs.append(torch.empty((b, eng.embed_dim), device=dev, dtype=torch.float32))
    torch.cuda.synchronize()
    per = B // b            # sub-batches per 256-utterance step
    def step(k):
        for j in range(per):
            i = (k * per + j) % nst
            with torch.cuda.stream(streams[i]):
                engs[i].embed_wave(wavs[k % 8][j * b:(j + 1) * b], out=outs[i], async_=True)
    t = run(step)
    print("%d engines, sub-batch %d: %.3f ms per 256 utterances, %.0f utt/s" % (nst, b, t * 1e3, B / t), flush=True)
    for e in engs: e.close()
