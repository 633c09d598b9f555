# PMC passes over the default bench (ECAPA bf16, B = 256) for per-kernel pipe counters.  bash tools/pmc_kernel.sh <tag> [bench args]
set -e
TAG=${1:-r03_pipe}
shift || true
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-scoring --no-extras $@"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq -o p -- $B > /dev/null 2> $OUT/sq.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES SQ_WAIT_INST_LDS --output-format csv -d $OUT/insts -o p -- $B > /dev/null 2> $OUT/insts.err
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/grbm -o p -- $B > /dev/null 2> $OUT/grbm.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- $B > /dev/null 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- $B > /dev/null 2> $OUT/write.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o p -- $B > /dev/null 2> $OUT/stats.err
ls $OUT
