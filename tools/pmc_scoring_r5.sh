# PMC passes over the scoring kernels at BASELINE configs[3] size (each pass its own run, --kernel-trace only).  bash tools/pmc_scoring_r5.sh <tag>
set -e
TAG=${1:-r05_scoring_pmc}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P="python3 $R/tools/score_prof.py"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq -o p -- $P > $OUT/sq.log 2> $OUT/sq.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- $P > $OUT/fetch.log 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- $P > $OUT/write.log 2> $OUT/write.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o p -- $P > $OUT/stats.log 2> $OUT/stats.err
python3 $R/tools/pmc_scoring_table.py $OUT | tee $OUT/table.txt
