# developer helper (round 4): F32X3 tests + the latency record of the default bench in one GPU call
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_ecapa.py tests/test_gpu_e2e.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r4_lat_tests.log 2>&1
tail -3 gpurun_out/r4_lat_tests.log
python bench.py --no-cpu-baseline --no-scoring --steps 10 --warmup 3 > gpurun_out/r4_lat_bench.json 2> gpurun_out/r4_lat_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4_lat_bench.json").read().strip().splitlines()[-1])
print("ecapa", d["value"], d["ms_per_step"], "f32x3", d.get("ecapa_f32x3", {}).get("value"))
for mode, rows in d["latency"].items():
    if not isinstance(rows, dict): continue
    for b, r in rows.items():
        if isinstance(r, dict) and "ms_per_call" in r:
            print(mode, b, round(r["ms_per_call"], 3), round(r["ms_per_call_pipelined"], 3), r["launches_per_call"], r["kernels"])
PY
