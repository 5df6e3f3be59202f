#!/bin/bash
# On the GPU box: the same bench command under several "ENV... LIB" configurations, alternating, with the single-pool trace split.
# usage: gpurun -- bash scripts/gpu_ab_cfg.sh "<env assignments> <lib.so>" ... -- [bench args...]
cd $GRAFT_REPO_ROOT
CFGS=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do CFGS+=("$1"); shift; done
shift
for i in 1 2; do
  for C in "${CFGS[@]}"; do
    L=${C##* }; E=${C% *}; [ "$E" = "$C" ] && E=""
    echo "== $C"
    env $E CGPT_LIB_PATH=$GRAFT_REPO_ROOT/cpugpupathtracing_amd/lib/$L timeout -k 10 300 python bench.py --cpu-seconds 0 --steps 3 "$@" 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print(d['value'], 'Mrays/s', d['ms_per_step'], 'ms/step | trace excl', r.get('kernel_ms_per_step'), 'round0', r.get('trace_ms_round0'), 'later', r.get('trace_ms_later'), '| excl pass', r.get('exclusive_pass_ms_per_step'), '| waves/SIMD', r.get('waves_per_simd'))
"
  done
done
