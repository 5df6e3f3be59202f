#!/bin/bash
# On the 1-GPU box: time chosen ranks' shares of an 8-GPU job, once per env setting given as an argument.
cd $GRAFT_REPO_ROOT
RANKS=${RANKS:-"7 0 3"}
for cfg in "$@"; do
  echo "== $cfg"
  for r in $RANKS; do
    env $cfg python bench.py --steps 4 --warmup 1 --cpu-seconds 0 --simulate-rank $r --simulate-world 8 $BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rank $r: ms_per_step', d['ms_per_step'], 'rays', d['config']['rays_per_step'], 'rows', d['config']['rows_per_gpu'])"
  done
done
