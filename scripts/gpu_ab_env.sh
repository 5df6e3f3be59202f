#!/bin/bash
# On the GPU box: the same bench command under two (or more) environment settings, alternating (A B A B), to separate a knob's effect from
# run-to-run variance.  usage: gpurun -- bash scripts/gpu_ab_env.sh "CGPT_WF_LDS_TRIS=0" "CGPT_WF_LDS_TRIS=1" -- [bench args...]
cd $GRAFT_REPO_ROOT
SETS=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do SETS+=("$1"); shift; done
shift
for i in 1 2; do
  for S in "${SETS[@]}"; do
    echo "== $S"
    env $S timeout -k 10 300 python bench.py --cpu-seconds 0 --steps 3 "$@" 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print(d['value'], 'Mrays/s', d['ms_per_step'], 'ms/step | trace excl', r.get('kernel_ms_per_step'), 'round0', r.get('trace_ms_round0'), 'later', r.get('trace_ms_later'), '| excl pass', r.get('exclusive_pass_ms_per_step'))
"
  done
done
