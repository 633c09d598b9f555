# PMC pass over the fused AS-norm kernel at BASELINE config 4's size (matrix-pipe busy, LDS activity, clock).  bash tools/pmc_asnorm.sh <tag>
set -e
TAG=${1:-r03_asnorm_pmc}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq -o p -- python3 $R/tools/score_prof.py > $OUT/sq.log 2> $OUT/sq.err
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/grbm -o p -- python3 $R/tools/score_prof.py > $OUT/grbm.log 2> $OUT/grbm.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o p -- python3 $R/tools/score_prof.py > $OUT/stats.log 2> $OUT/stats.err
find $OUT -name "*.csv" | head
