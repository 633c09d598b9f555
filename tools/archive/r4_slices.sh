# developer helper (round 4): Res2Net chain time slices at B = 256 (SVHIP_R2_SLICES read at svhip_create), one GPU call
cd $GRAFT_REPO_ROOT
for S in -1 2 3 -1 2; do
  export SVHIP_R2_SLICES=$S
  python bench.py --no-cpu-baseline --no-scoring --no-extras --steps 20 --warmup 5 > gpurun_out/sl.json 2> gpurun_out/sl.err
  S=$S python - <<'PY'
import json, os
d = json.loads(open("gpurun_out/sl.json").read().strip().splitlines()[-1])
print("slices", os.environ["S"], round(d["value"]), round(d["ms_per_step"], 3), d["check"]["ok"],
      " ".join("%s=%.1f" % (k, x["avg_ms"] * 1e3) for k, x in d["kernels"].items() if "res2" in k or k in ("gemm_pw3", "se_apply")))
PY
done
