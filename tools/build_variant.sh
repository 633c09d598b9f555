#!/bin/bash
# developer tool: build a second libsvhip (tools/libsvhip_var.so) from the working tree with extra -D flags, for same-box A/B:
#   tools/build_variant.sh -DQGROUP_M_OVERRIDE=8 ;  SVHIP_LIB_PATH=$PWD/tools/libsvhip_var.so python bench.py ...
set -e
cd "$(dirname "$0")/.."
CS=speakerverification_amd/csrc
mkdir -p tools/_var
for f in $CS/*.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -ffp-contract=on "$@" -c $f -o tools/_var/$(basename ${f%.hip}).o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 tools/_var/*.o -o tools/libsvhip_var.so
