# PMC passes (each in its own run, --kernel-trace only): HBM/fabric traffic and fabric read latency per kernel.  bash tools/pmc_round2.sh <tag>
set -e
TAG=${1:-r02_pmc}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-scoring --no-extras"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/ecapa_fetch -o p -- $B > /dev/null 2> $OUT/ecapa_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/ecapa_write -o p -- $B > /dev/null 2> $OUT/ecapa_write.err
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $OUT/ecapa_lat -o p -- $B > /dev/null 2> $OUT/ecapa_lat.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/rawnet2_fetch -o p -- $B --model rawnet2 > /dev/null 2> $OUT/rawnet2_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/rawnet2_write -o p -- $B --model rawnet2 > /dev/null 2> $OUT/rawnet2_write.err
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $OUT/rawnet2_lat -o p -- $B --model rawnet2 > /dev/null 2> $OUT/rawnet2_lat.err
ls $OUT/*/ | head -30
