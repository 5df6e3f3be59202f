#!/bin/bash
# On the GPU box: step-fill statistics of the trace kernel (CGPT_WF_PROFILE) for each env setting given as an argument.
cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  echo "== $cfg"
  env CGPT_WF_PROFILE=1 $cfg python bench.py --steps 2 --warmup 1 --kernel wavefront --cpu-seconds 0 2>&1 | grep -v amdgpu.ids | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'])
    else: print(l.rstrip())"
done
