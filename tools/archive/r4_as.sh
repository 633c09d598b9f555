# developer helper (round 4): scoring tests + AS-norm timing, 16-wide against 32-wide kernel, one GPU call
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_scoring.py -x -q -s > gpurun_out/r4_as_tests.log 2>&1
grep -E "passed|failed|16-wide|Error" gpurun_out/r4_as_tests.log | tail -6
python - <<'PY'
import time, torch
from speakerverification_amd.engine import Engine
dev = torch.device("cuda:0")
eng = Engine(model="none", max_batch=1)
g = torch.Generator(device=dev); g.manual_seed(1)
N, D, K, top = 1_200_000, 192, 5994, 200
E = torch.randn((N, D), generator=g, device=dev); eng.l2norm_(E)
C = torch.randn((K, D), generator=g, device=dev); eng.l2norm_(C)
for rep in range(2):
    for w32 in (1, 0):
        eng.set_option("asnorm_w32", w32)
        eng.asnorm_stats(E, C, top)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): mu, sd = eng.asnorm_stats(E, C, top)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print("asnorm_w32 =", w32, ": %.2f ms per 1.2 M embeddings" % (dt * 1e3), "fallback", eng.asnorm_last_fallback, flush=True)
PY
