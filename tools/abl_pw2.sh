timeout -k 10 120 tools/gemm_bench 1 256 0 1 1 | grep -E "pw2|tdnn|mfa|blk0"
