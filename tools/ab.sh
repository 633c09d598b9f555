# developer helper: A/B two builds of the library in ONE GPU call (box-to-box variance is +-3 %).  tools/_ab/libsvhip_A.so vs the in-tree build.
#   bash tools/ab.sh [ecapa|rawnet2]
cd $GRAFT_REPO_ROOT
M=${1:-ecapa}
EXTRA=""
if [ "$M" = "rawnet2" ]; then EXTRA="--model rawnet2 --compute f16"; fi
for rep in 1 2; do
  for v in A B; do
    if [ $v = A ]; then export SVHIP_LIB_PATH=$GRAFT_REPO_ROOT/tools/_ab/libsvhip_A.so; else unset SVHIP_LIB_PATH; fi
    python bench.py $EXTRA --no-cpu-baseline --no-scoring --no-extras --steps 20 --warmup 5 > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err
    V=$v python - <<'PY'
import json, os
v = os.environ["V"]
d = json.loads(open("gpurun_out/ab_%s.json" % v).read().strip().splitlines()[-1])
print(v, round(d["value"]), round(d["ms_per_step"], 3), "sustained", round(d["sustained"]["value"]), d["check"]["ok"],
      " ".join("%s=%.1f" % (k, x["avg_ms"] * 1e3) for k, x in d["kernels"].items() if x["ms_per_step"] > 0.1))
PY
  done
done
