# one GPU call: every number DESIGN.md / profiles/ quote for this round.   bash tools/snapshot_round4.sh <tag>
set -e
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench done"
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --no-scoring --no-extras > $OUT/bench_launcher_w1.json 2> $OUT/bench_launcher_w1.err
python3 bench.py --config shard > $OUT/bench_shard_125k.json 2> $OUT/bench_shard.err
python3 bench.py --model rawnet2 --compute f16 --steps 30 --warmup 5 --no-cpu-baseline --no-scoring --no-extras > $OUT/bench_rawnet2.json 2> $OUT/bench_rawnet2.err
echo "benches done"
bash tools/profile_round4.sh $TAG/prof > $OUT/profile.log 2>&1
echo "profiles done"
bash tools/pmc_round4.sh $TAG/pmc > $OUT/pmc.log 2>&1
python3 tools/pmc_summary3.py $OUT/pmc $OUT/pmc_summary.json $OUT/pmc_traffic.json > $OUT/pmc_table.txt
python3 - <<PY
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("ECAPA", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["check"]["ok"], "sustained", d["sustained"]["value"])
for k in ("rawnet2","rawnet2_bf16","ecapa_f32","ecapa_f32x3","rawnet2_f32x3","fusion","pcie"): print(k, d.get(k,{}).get("value"), d.get(k,{}).get("ms_per_step"))
s=json.loads(open("$OUT/bench_shard_125k.json").read().strip().splitlines()[-1]); print("shard", s["value"], s.get("end_to_end_seconds"))
PY
