# developer helper (round 4): headline + rawnet2 benches without extras; per-kernel tables
cd $GRAFT_REPO_ROOT
bash tools/r4_kern.sh
python bench.py --model rawnet2 --compute f16 --no-cpu-baseline --no-scoring --no-extras --steps 20 --warmup 5 > gpurun_out/r4_kern_rn.json 2> gpurun_out/r4_kern_rn.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4_kern_rn.json").read().strip().splitlines()[-1])
print("rawnet2", round(d["value"]), round(d["ms_per_step"], 3), d["check"]["ok"])
for k, v in d["kernels"].items():
    print("  %-16s %7.1f us x %d = %7.1f" % (k, v["avg_ms"] * 1e3, v["launches_per_step"], v["ms_per_step"] * 1e3))
PY
