for ds in 1 0; do for dbg in 0 512; do echo "== dscale $ds debug $dbg"; timeout -k 10 60 tools/gemm_bench 1 256 $dbg $ds | grep -E "tdnn  N1024 K1024 gelu|mfa" ; done; done
