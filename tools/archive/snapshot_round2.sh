# one GPU call: every number DESIGN.md / profiles/ quote for this round.   bash tools/snapshot_round2.sh <tag>
set -e
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python3 bench.py --steps 30 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 3 > $OUT/bench_launcher_w1.json 2> $OUT/bench_launcher_w1.err
python3 bench.py --config shard > $OUT/bench_shard_125k.json 2> $OUT/bench_shard.err
python3 bench.py --model rawnet2 --steps 30 --warmup 5 --no-cpu-baseline --no-scoring > $OUT/bench_rawnet2.json 2> $OUT/bench_rawnet2.err
bash tools/profile_round2.sh $TAG/prof > $OUT/profile.log 2>&1
bash tools/pmc_round2.sh $TAG/pmc > $OUT/pmc.log 2>&1
python3 tools/pmc_summary2.py $OUT/pmc $OUT/pmc_summary.json > /dev/null
python3 - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print("ECAPA", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["check"]["ok"])
for k in ("rawnet2","ecapa_f32","pcie"): print(k, d[k].get("value"), d[k].get("ms_per_step"))
print("W1", json.load(open("$OUT/bench_launcher_w1.json"))["shard"]["allgather_ms"])
s=json.load(open("$OUT/bench_shard_125k.json")); print("shard", s["value"], s["end_to_end_seconds"])
PY
