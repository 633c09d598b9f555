#!/bin/bash
# developer tool: build tools/gemm_bench with the GEMM sources compiled with -DSVHIP_GEMM_DEBUG (ablation flags)
set -e
cd "$(dirname "$0")/.."
CS=speakerverification_amd/csrc
FL="-O3 -std=c++17 --offload-arch=gfx950 -DSVHIP_GEMM_DEBUG -I $CS"
for f in gemm gemm_pw gemm_pw2 gemm_pw3 gemm_n128; do /opt/rocm/bin/hipcc $FL -c $CS/$f.hip -o tools/$f.dbg.o & done
/opt/rocm/bin/hipcc $FL -c tools/gemm_bench.hip -o tools/gemm_bench.o
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 tools/gemm_bench.o tools/gemm.dbg.o tools/gemm_pw.dbg.o tools/gemm_pw2.dbg.o tools/gemm_pw3.dbg.o tools/gemm_n128.dbg.o -o tools/gemm_bench
exit 0
/opt/rocm/bin/hipcc $FL -c tools/r2_bench.hip -o tools/r2_bench.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 tools/r2_bench.o tools/gemm_pw2.dbg.o tools/gemm_pw3.dbg.o -o tools/r2_bench
/opt/rocm/bin/hipcc $FL -c $CS/res2net.hip -o tools/res2net.dbg.o
/opt/rocm/bin/hipcc $FL -c tools/res2_bench.hip -o tools/res2_bench.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 tools/res2_bench.o tools/res2net.dbg.o -o tools/res2_bench
/opt/rocm/bin/hipcc $FL -c $CS/asp_fused.hip -o tools/asp_fused.dbg.o
/opt/rocm/bin/hipcc $FL -c tools/asp_bench.hip -o tools/asp_bench.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 tools/asp_bench.o tools/asp_fused.dbg.o -o tools/asp_bench
