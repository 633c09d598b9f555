#!/bin/bash
# per-kernel VGPR / scratch table of one csrc file: tools/regs.sh gemm_pw3.hip [extra hipcc flags]
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -ffp-contract=on -Rpass-analysis=kernel-resource-usage "$@" \
  -c /root/repo/speakerverification_amd/csrc/$f -o /tmp/regs_$$.o 2>&1 | grep -E "Function Name|  VGPRs:|ScratchSize" \
  | sed 's/.*remark: *//; s/\[-Rpass.*//' | paste - - - | sed 's/Function Name: //; s/ScratchSize \[bytes\/lane\]/scratch/' 
rm -f /tmp/regs_$$.o
