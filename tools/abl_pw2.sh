timeout -k 10 120 tools/gemm_bench 1 256 0,4096,128,4224 1 3 | grep -E "tdnn  N1024 K1024 gelu|blk0|none"
