#!/bin/bash
# developer tool (round 6): same-box, same-process yardstick for gemm_pw3 — the vendor's plain bf16 GEMM (hipBLASLt,
# best of its heuristic candidates) against gemm_pw3 with its epilogue on and ablated (PW3_ABL: 4 no activation, 8 no output stores),
# on random operands, for [102 656 x 1024].[1024 x 1024]^T and [102 656 x 3072].[3072 x 3072]^T.
#   build (here):  bash tools/pw3_yardstick.sh build      run (GPU box):  bash tools/pw3_yardstick.sh > gpurun_out/r06_pw3_yardstick.txt
set -e
cd "$(dirname "$0")/.."
CS=speakerverification_amd/csrc
FL="-O3 -std=c++17 --offload-arch=gfx950 -DSVHIP_GEMM_DEBUG -I $CS"
VARS="${VARS:-0 4 12}"
if [ "$1" = "build" ]; then
  for f in gemm gemm_pw gemm_pw2 gemm_n128; do /opt/rocm/bin/hipcc $FL -c $CS/$f.hip -o tools/$f.dbg.o & done
  /opt/rocm/bin/hipcc $FL -DGEMM_BENCH_VENDOR -c tools/gemm_bench.hip -o tools/gemm_bench_v.o &
  for v in $VARS; do /opt/rocm/bin/hipcc $FL -DPW3_ABL=$v -c $CS/gemm_pw3.hip -o tools/gemm_pw3.abl$v.o & done
  wait
  for v in $VARS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 tools/gemm_bench_v.o tools/gemm.dbg.o tools/gemm_pw.dbg.o tools/gemm_pw2.dbg.o tools/gemm_n128.dbg.o tools/gemm_pw3.abl$v.o -L/opt/rocm/lib -lhipblaslt -o tools/gemm_bench_y$v
  done
  exit 0
fi
ROUNDS=${ROUNDS:-4}
for r in $(seq 1 $ROUNDS); do
  for v in $VARS; do
    echo "== round $r  PW3_ABL=$v  (columns: shape | variant | ms | TFLOP/s)"
    timeout -k 10 180 tools/gemm_bench_y$v 1 256 0,1048576 1 2 | grep -E "tdnn  N1024 K1024 gelu|mfa|tdnn  N1024 K1024 none" | awk '{print $1, $2, $3, $4, "|", $5, $6, "|", $7, $8, "|", $9, $10}'
  done
done
