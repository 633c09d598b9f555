"""BASELINE configs[3] under rocprofv3 (tools/profile_round2.sh): one pass of every scoring kernel at full size."""
import sys, time
sys.path.insert(0, '.')
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speakerverification_amd.engine import Engine
dev = torch.device("cuda", 0)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    eng = Engine(model="none", device=0, stream=torch.cuda.current_stream().cuda_stream)
    N, K, P = 1_200_000, 5994, 1_200_000
    g = torch.Generator(device=dev).manual_seed(2)
    E = torch.randn((N, 192), generator=g, device=dev); eng.l2norm_(E)
    C = torch.randn((K, 192), generator=g, device=dev); eng.l2norm_(C)
    ia = torch.arange(P, device=dev, dtype=torch.int32)
    ib = torch.randperm(N, generator=g, device=dev)[:P].to(torch.int32)
    out = torch.empty(P, device=dev)
    eng.asnorm_stats(E[:100000], C, 200)
    eng.score_pairs(E, ia, ib, out)
    for rep in range(3):          # end to end, no per-kernel events
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.asnorm_stats(E, C, 200)
        torch.cuda.synchronize(); print("asnorm_stats end to end %.2f ms (fallback %d)" % ((time.perf_counter() - t0) * 1e3, eng.asnorm_last_fallback))
    eng.profile(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mu, sd = eng.asnorm_stats(E, C, 200)
    for _ in range(3):
        eng.score_pairs(E, ia, ib, out)
        eng.asnorm_pairs(E, mu, sd, ia, ib, out)
    dense = torch.empty((16384, 16384), device=dev)
    for _ in range(3):
        eng.score_matrix(E[:16384], E[16384:32768], dense)
    torch.cuda.synchronize(); print("total", time.perf_counter() - t0)
    for k, v in eng.profile_results().items(): print(k, v)
