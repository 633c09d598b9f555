# rocprofv3 evidence for every configuration (kernel stats + kernel trace); run on the GPU box:  bash tools/profile_round4.sh <tag>
set -e
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-scoring --no-extras --sustain-seconds 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ecapa_bf16 -o p -- python3 $R/bench.py --steps 10 --warmup 3 $B > $OUT/ecapa_bf16_bench.json 2> $OUT/ecapa_bf16.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rawnet2_f16 -o p -- python3 $R/bench.py --model rawnet2 --compute f16 --steps 10 --warmup 3 $B > $OUT/rawnet2_f16_bench.json 2> $OUT/rawnet2_f16.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ecapa_f32 -o p -- python3 $R/bench.py --compute f32 --steps 3 --warmup 1 $B > $OUT/ecapa_f32_bench.json 2> $OUT/ecapa_f32.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ecapa_f32x3 -o p -- python3 $R/bench.py --compute f32x3 --steps 4 --warmup 1 $B > $OUT/ecapa_f32x3_bench.json 2> $OUT/ecapa_f32x3.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rawnet2_f32x3 -o p -- python3 $R/bench.py --model rawnet2 --compute f32x3 --steps 4 --warmup 1 $B > $OUT/rawnet2_f32x3_bench.json 2> $OUT/rawnet2_f32x3.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/scoring -o p -- python3 $R/tools/score_prof.py > $OUT/scoring.log 2> $OUT/scoring.err
cd $R
for d in ecapa_bf16 rawnet2_f16 ecapa_f32 ecapa_f32x3 rawnet2_f32x3 scoring; do f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); echo "== $d"; head -12 $f | cut -c1-160; done
