#!/bin/bash
# On the GPU box: ms per one-sample cgpt_render call (scripts/gpu_small_calls.py) with several builds of the library, alternating.
# usage: gpurun -- bash scripts/gpu_one_sample_ab.sh <libA.so> <libB.so> ... -- [gpu_small_calls.py args]
cd $GRAFT_REPO_ROOT
LIBS=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do LIBS+=("$1"); shift; done
shift
for i in 1 2; do
  for L in "${LIBS[@]}"; do
    echo "== $L"
    CGPT_LIB_PATH=$GRAFT_REPO_ROOT/cpugpupathtracing_amd/lib/$L timeout -k 10 200 python scripts/gpu_small_calls.py "$@" 2>/dev/null | grep -v "^#"
  done
done
