#!/bin/bash
# developer tool: dense 16 384^2 x 192 score matrix, the in-tree library against tools/libsvhip_var.so, alternated in one GPU call
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for v in var tree; do
    if [ $v = var ]; then export SVHIP_LIB_PATH=$GRAFT_REPO_ROOT/tools/libsvhip_var.so; else unset SVHIP_LIB_PATH; fi
    echo "== $v"; python tools/dense_probe.py 2>&1 | grep -E "dense|err"
  done
done
