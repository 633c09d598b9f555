#!/bin/bash
# developer tool: one gemm_bench per compile-time ablation of gemm_pw3.hip (PW3_ABL bits: 1 no MFMAs, 2 no operand DMAs, 4 no
# activation, 8 no output stores, 16 no fragment reads, 64 no column sums, 128 strict vmcnt).
#   build:  bash tools/abl_pw3.sh build     run (GPU box):  bash tools/abl_pw3.sh
set -e
cd "$(dirname "$0")/.."
CS=speakerverification_amd/csrc
FL="-O3 -std=c++17 --offload-arch=gfx950 -DSVHIP_GEMM_DEBUG -I $CS"
VARS="${VARS:-0 1 2 4 16 128}"
DBGS="${DBGS:-0,4096,384}"
if [ "$1" = "build" ]; then
  for v in $VARS; do
    /opt/rocm/bin/hipcc $FL -DPW3_ABL=$v -c $CS/gemm_pw3.hip -o tools/gemm_pw3.abl$v.o &
  done
  wait
  for v in $VARS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 tools/gemm_bench.o tools/gemm.dbg.o tools/gemm_pw.dbg.o tools/gemm_pw2.dbg.o tools/gemm_pw3.abl$v.o -o tools/gemm_bench_abl$v
  done
  exit 0
fi
for v in $VARS; do
  echo "== PW3_ABL=$v"
  timeout -k 10 120 tools/gemm_bench_abl$v 1 256 $DBGS 1 2 | grep -E "tdnn  N1024 K1024 gelu|mfa" | awk '{print $1, $2, $3, $4, $5, $6, $7, $8, $9, $13}'
done
