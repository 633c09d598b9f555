#!/bin/bash
# developer tool (round 6): tools/gemm_bench_w4 = gemm_bench + the four-wave prototype (debug bit 2097152) + the vendor yardstick (1048576)
set -e
cd "$(dirname "$0")/.."
CS=speakerverification_amd/csrc
FL="-O3 -std=c++17 --offload-arch=gfx950 -DSVHIP_GEMM_DEBUG -I $CS"
for f in gemm gemm_pw gemm_pw2 gemm_pw3 gemm_n128 gemm_pw4; do
  if [ ! -f tools/$f.dbg.o ] || [ $CS/$f.hip -nt tools/$f.dbg.o ] || [ -n "$FORCE" ]; then /opt/rocm/bin/hipcc $FL $EXTRA -c $CS/$f.hip -o tools/$f.dbg.o & fi
done
/opt/rocm/bin/hipcc $FL -DGEMM_BENCH_VENDOR -c tools/gemm_bench.hip -o tools/gemm_bench_v.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 tools/gemm_bench_v.o tools/gemm.dbg.o tools/gemm_pw.dbg.o tools/gemm_pw2.dbg.o tools/gemm_pw3.dbg.o tools/gemm_n128.dbg.o tools/gemm_pw4.dbg.o -L/opt/rocm/lib -lhipblaslt -o tools/gemm_bench_w4
