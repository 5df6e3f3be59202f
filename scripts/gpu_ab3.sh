#!/bin/bash
# On the GPU box: GPU tests of the default build, then the same bench command with several builds of the library, alternating, with the
# single-pool trace split.  usage: gpurun -- bash scripts/gpu_ab3.sh "<pytest args or ->" <libA.so> <libB.so> ... -- [bench args...]
cd $GRAFT_REPO_ROOT
T=$1; shift
LIBS=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do LIBS+=("$1"); shift; done
shift
if [ "$T" != "-" ]; then timeout -k 10 600 python -m pytest $T -m gpu -x -q 2>&1 | tail -6 || exit 1; fi
for i in 1 2; do
  for L in "${LIBS[@]}"; do
    echo "== $L"
    CGPT_LIB_PATH=$GRAFT_REPO_ROOT/cpugpupathtracing_amd/lib/$L timeout -k 10 300 python bench.py --cpu-seconds 0 --steps 3 "$@" 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print(d['value'], 'Mrays/s', d['ms_per_step'], 'ms/step | trace excl', r.get('kernel_ms_per_step'), 'round0', r.get('trace_ms_round0'), 'later', r.get('trace_ms_later'), '| excl pass', r.get('exclusive_pass_ms_per_step'))
"
  done
done
