import sys, time
sys.path.insert(0, '.')
import torch
from speakerverification_amd.engine import Engine
dev = torch.device("cuda", 0)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    eng = Engine(model="none", device=0, stream=torch.cuda.current_stream().cuda_stream)
    N, K = 1_200_000, 5994
    g = torch.Generator(device=dev).manual_seed(2)
    E = torch.randn((N, 192), generator=g, device=dev); eng.l2norm_(E)
    C = torch.randn((K, 192), generator=g, device=dev); eng.l2norm_(C)
    eng.asnorm_stats(E[:100000], C, 200)
    eng.profile(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mu, sd = eng.asnorm_stats(E, C, 200)
    torch.cuda.synchronize(); print("asnorm_stats total", time.perf_counter() - t0)
    for k, v in eng.profile_results().items(): print(k, v)
