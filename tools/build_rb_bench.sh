#!/bin/bash
# developer tool: build tools/rb_bench (fused RawNet2 block kernel with phase stamps)
set -e
cd "$(dirname "$0")/.."
CS=speakerverification_amd/csrc
FL="-O3 -std=c++17 --offload-arch=gfx950 -DSVHIP_GEMM_DEBUG -I $CS -fno-fast-math -ffp-contract=on"
/opt/rocm/bin/hipcc $FL -c $CS/rn_block128.hip -o tools/rn_block128.dbg.o
/opt/rocm/bin/hipcc $FL -c tools/rb_bench.hip -o tools/rb_bench.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 tools/rb_bench.o tools/rn_block128.dbg.o -o tools/rb_bench
