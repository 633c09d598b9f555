# developer helper (round 4): the headline bench without extras; prints the per-kernel table
cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline --no-scoring --no-extras --steps 20 --warmup 5 > gpurun_out/r4_kern.json 2> gpurun_out/r4_kern.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4_kern.json").read().strip().splitlines()[-1])
print("ecapa", round(d["value"]), round(d["ms_per_step"], 3), "sustained", round(d["sustained"]["value"]), d["check"]["ok"])
for k, v in d["kernels"].items():
    print("  %-16s %7.1f us x %d = %7.1f" % (k, v["avg_ms"] * 1e3, v["launches_per_step"], v["ms_per_step"] * 1e3))
PY
