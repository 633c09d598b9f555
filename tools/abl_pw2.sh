timeout -k 10 120 tools/gemm_bench 1 256 0,512 1 3 | grep -E "blk0"
