#!/bin/bash
# On the GPU box: the same bench command with two builds of the library, alternating (A B A B), to separate a code change from
# box-to-box and run-to-run variance.  usage: gpurun -- bash scripts/gpu_ab_libs.sh <libA.so> <libB.so> [bench args...]
cd $GRAFT_REPO_ROOT
A=$1; B=$2; shift 2
for i in 1 2; do
  for L in $A $B; do
    echo "== $L"
    CGPT_LIB_PATH=$GRAFT_REPO_ROOT/$L timeout -k 10 300 python bench.py --cpu-seconds 0 --no-roofline-pass --steps 3 "$@" 2>/dev/null | cut -c70-200
  done
done
