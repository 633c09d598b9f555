# developer helper (round 4): x3 fuzz distribution + scoring tests + f32x3 bench in one GPU call
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python tools/x3_fuzz.py 120 7 > gpurun_out/r4_x3h_fuzz.txt 2>&1
tail -2 gpurun_out/r4_x3h_fuzz.txt
python - <<'PY'
import re, statistics
v = sorted(float(m.group(1)) for m in re.finditer(r"rel err ([0-9.e+-]+)", open("gpurun_out/r4_x3h_fuzz.txt").read()))
print("x3 fuzz: n", len(v), "median", statistics.median(v), "p99", v[int(len(v) * 0.99) - 1], "max", v[-1])
PY
timeout -k 10 600 python -m pytest tests/test_gpu_scoring.py -x -q -s > gpurun_out/r4_scoring_tests.log 2>&1
grep -E "passed|failed|h3|x6|x3 mu" gpurun_out/r4_scoring_tests.log | tail -8
python bench.py --compute f32x3 --no-cpu-baseline --no-extras --steps 6 --warmup 2 > gpurun_out/r4_x3h_bench.json 2> gpurun_out/r4_x3h_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4_x3h_bench.json").read().strip().splitlines()[-1])
print("f32x3", d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["check"])
s = d["scoring"]
print("scoring: asnorm_stats_s", s["asnorm_stats_s"], "pairs/s", s["asnorm_pairs_per_s"], "TF", s["asnorm_cohort_gemm_TFLOPs"], "dense TF", s["dense_TFLOPs"], "cos pairs/s", s["cosine_pairs_per_s"])
PY
