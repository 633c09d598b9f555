# PMC passes (each in its own run, --kernel-trace only): HBM/fabric traffic and fabric read latency per kernel.  bash tools/pmc_round3.sh <tag>
set -e
TAG=${1:-r03_pmc}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-scoring --no-extras"
LAT="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum"
for m in ecapa rawnet2 f32x3; do
  case $m in ecapa) X="";; rawnet2) X="--model rawnet2";; f32x3) X="--compute f32x3";; esac
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${m}_fetch -o p -- $B $X > /dev/null 2> $OUT/${m}_fetch.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${m}_write -o p -- $B $X > /dev/null 2> $OUT/${m}_write.err
  rocprofv3 --kernel-trace --pmc $LAT --output-format csv -d $OUT/${m}_lat -o p -- $B $X > /dev/null 2> $OUT/${m}_lat.err
  echo "pmc $m done"
done
