# developer tool (round 6): effective clock and matrix-pipe busy share per gemm_pw3 instance INSIDE the model.
#   bash tools/pmc_clock_r6.sh <tag> [bench args]      (GPU box; the program stands directly behind "--")
# Pass 1: GRBM_GUI_ACTIVE (sum over the 8 XCDs: / 8 / kernel wall = effective clock, MI355X_MICROARCH.md "DVFS give-back").
# Pass 2: SQ_BUSY_CYCLES, SQ_VALU_MFMA_BUSY_CYCLES, SQ_WAVE_CYCLES, SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY, SQ_INSTS_MFMA, SQ_BUSY_CU_CYCLES.
set -e
TAG=${1:-r06_clock}
shift || true
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-scoring --no-extras --sustain-seconds 0 $@"
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/grbm -o p -- $B > /dev/null 2> $OUT/grbm.err
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/sq -o p -- $B > /dev/null 2> $OUT/sq.err
python3 $R/tools/pmc_clock_table.py $OUT
