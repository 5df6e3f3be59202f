#!/bin/bash
# On the 1-GPU box: time each rank's share of an N-GPU job separately (no collective) to see how balanced the bands are.
cd $GRAFT_REPO_ROOT
for N in ${@:-8}; do
  for r in $(seq 0 $((N-1))); do
    python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --simulate-rank $r --simulate-world $N 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rank $r of $N: ms_per_step', d['ms_per_step'], 'rays', d['config']['rays_per_step'], 'Mrays/s', d['value'])"
  done
done
