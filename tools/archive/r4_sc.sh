# developer helper (round 4): scoring tests + dense score matrix timing, row-streaming against tiled
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_scoring.py -x -q -s > gpurun_out/r4_sc_tests.log 2>&1
grep -E "passed|failed|score_matrix [0-9]|Error|assert" gpurun_out/r4_sc_tests.log | tail -12
python - <<'PY'
import time, torch
from speakerverification_amd.engine import Engine
dev = torch.device("cuda:0")
eng = Engine(model="none", max_batch=1)
g = torch.Generator(device=dev); g.manual_seed(1)
for n in (16384, 8192):
    A = torch.randn((n, 192), generator=g, device=dev); eng.l2norm_(A)
    B = torch.randn((n, 192), generator=g, device=dev); eng.l2norm_(B)
    out = torch.empty((n, n), device=dev)
    for tiled in (1, 0, 1, 0):
        eng.set_option("score_tiled", tiled)
        eng.score_matrix(A, B, out); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): eng.score_matrix(A, B, out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print("n", n, "score_tiled =", tiled, ": %.3f ms, %.0f TFLOP/s, %.2f TB/s of output" % (dt * 1e3, 2.0 * n * n * 192 / dt / 1e12, 4.0 * n * n / dt / 1e12), flush=True)
PY
