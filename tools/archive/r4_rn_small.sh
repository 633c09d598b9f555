# developer helper (round 4): RawNet2 tests + the latency record
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_rawnet2.py -x -q -m gpu -s > gpurun_out/r4_rn_tests.log 2>&1
grep -E "passed|failed|sliced vs|Error|error" gpurun_out/r4_rn_tests.log | tail -12
bash tools/r4_lat2.sh 2>&1 | grep rawnet2
