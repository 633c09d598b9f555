set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -o stats -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-scoring > $R/gpurun_out/bench_under_rocprof.json 2> $R/gpurun_out/prof_stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -o fetch -- python3 $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-scoring > /dev/null 2> $R/gpurun_out/prof_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -o write -- python3 $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-scoring > /dev/null 2> $R/gpurun_out/prof_write.err
cd $R
python3 tools/pmc_summary.py gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/pmc_traffic.json
find gpurun_out/prof_stats -name "*kernel_stats.csv" | head -2
python3 bench.py --steps 30 --warmup 5 > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
tail -c 600 gpurun_out/bench_under_rocprof.json
