#!/bin/bash
# developer tool: one gemm_bench per compile-time ablation of gemm_pw2.hip (PW2_ABL bits: 1 no MFMAs, 2 no operand DMAs, 4 no
# activation, 8 no output stores, 16 no fragment reads).   build:  bash tools/abl_pw2.sh build     run (GPU box):  bash tools/abl_pw2.sh
set -e
cd "$(dirname "$0")/.."
CS=speakerverification_amd/csrc
FL="-O3 -std=c++17 --offload-arch=gfx950 -DSVHIP_GEMM_DEBUG -I $CS"
VARS="${VARS:-0 1 2 4 8 16 32}"
if [ "$1" = "build" ]; then
  for v in $VARS; do
    /opt/rocm/bin/hipcc $FL -DPW2_ABL=$v -c $CS/gemm_pw2.hip -o tools/gemm_pw2.abl$v.o &
  done
  wait
  for v in $VARS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 tools/gemm_bench.o tools/gemm.dbg.o tools/gemm_pw.dbg.o tools/gemm_pw2.abl$v.o -o tools/gemm_bench_abl$v
  done
  exit 0
fi
for v in $VARS; do
  echo "== PW2_ABL=$v"
  timeout -k 10 120 tools/gemm_bench_abl$v 1 256 0 1 2 | grep -E "tdnn  N1024 K1024 gelu|mfa" | awk '{print $1, $2, $3, $4, $7, $8, $9, $10}'
done
