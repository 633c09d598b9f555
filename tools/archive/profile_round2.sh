# rocprofv3 evidence for every configuration (kernel stats + kernel trace); run on the GPU box:  bash tools/profile_round2.sh <tag>
set -e
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ecapa_bf16 -o p -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-scoring --no-extras > $OUT/ecapa_bf16_bench.json 2> $OUT/ecapa_bf16.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rawnet2_bf16 -o p -- python3 $R/bench.py --model rawnet2 --steps 10 --warmup 3 --no-cpu-baseline --no-scoring --no-extras > $OUT/rawnet2_bf16_bench.json 2> $OUT/rawnet2_bf16.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ecapa_f32 -o p -- python3 $R/bench.py --compute f32 --steps 3 --warmup 1 --no-cpu-baseline --no-scoring --no-extras > $OUT/ecapa_f32_bench.json 2> $OUT/ecapa_f32.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ecapa_f32x3 -o p -- python3 $R/bench.py --compute f32x3 --steps 4 --warmup 1 --no-cpu-baseline --no-scoring --no-extras > $OUT/ecapa_f32x3_bench.json 2> $OUT/ecapa_f32x3.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/scoring -o p -- python3 $R/tools/score_prof.py > $OUT/scoring.log 2> $OUT/scoring.err
cd $R
for d in ecapa_bf16 rawnet2_bf16 ecapa_f32 ecapa_f32x3 scoring; do f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); echo "== $d"; head -12 $f | cut -c1-160; done
