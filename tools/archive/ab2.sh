# developer helper: A/B (tools/_ab/libsvhip_A.so vs in-tree) on the bf16 headline AND the f32x3 path, one GPU call
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for v in A B; do
    if [ $v = A ]; then export SVHIP_LIB_PATH=$GRAFT_REPO_ROOT/tools/_ab/libsvhip_A.so; else unset SVHIP_LIB_PATH; fi
    for mode in bf16 f32x3; do
      python bench.py --compute $mode --no-cpu-baseline --no-scoring --no-extras --steps 10 --warmup 3 --sustain-seconds 0 > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err
      V=$v MODE=$mode python - <<'PY'
import json, os
v = os.environ["V"]
d = json.loads(open("gpurun_out/ab_%s.json" % v).read().strip().splitlines()[-1])
print(v, os.environ["MODE"], round(d["value"]), round(d["ms_per_step"], 3), d["check"]["ok"],
      " ".join("%s=%.1f" % (k, x["avg_ms"] * 1e3) for k, x in d["kernels"].items() if x["ms_per_step"] > 0.1))
PY
    done
  done
done
