#!/bin/bash
# On the GPU box: bench the wavefront path under CGPT_WF_* settings given one per line on stdin-like args, e.g.
#   gpurun -- bash scripts/gpu_sweep.sh "CGPT_WF_POOLS=4" "CGPT_WF_POOLS=8 CGPT_WF_BATCH=32"
cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg python bench.py --steps ${STEPS:-2} --warmup 1 --kernel wavefront --cpu-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
