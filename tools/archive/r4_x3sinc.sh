# developer helper (round 4): rawnet2 tests + f32x3 rawnet2 bench
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_rawnet2.py -x -q -m gpu -s > gpurun_out/r4_x3sinc_tests.log 2>&1
grep -E "passed|failed|split-half|Error|assert" gpurun_out/r4_x3sinc_tests.log | tail -10
for v in 1 0; do
SVHIP_RN_SINC_F32=$v python bench.py --model rawnet2 --compute f32x3 --no-cpu-baseline --no-scoring --no-extras --steps 6 --warmup 2 --sustain-seconds 0 > gpurun_out/x3s.json 2> gpurun_out/x3s.err
V=$v python - <<'PY'
import json, os
d = json.loads(open("gpurun_out/x3s.json").read().strip().splitlines()[-1])
print("rn_sinc_f32 =", os.environ["V"], round(d["value"]), round(d["ms_per_step"], 3), d["check"], " ".join("%s=%.1f" % (k, x["avg_ms"] * 1e3) for k, x in d["kernels"].items() if x["ms_per_step"] > 0.2))
PY
done
