#!/bin/bash
# On the GPU box: every rank's share of an 8-GPU split of the default frame for several interleaved band heights (the slowest share is the
# 8-GPU frame time).  usage: gpurun -- bash scripts/gpu_band_rows.sh [band rows ...]
cd $GRAFT_REPO_ROOT
for b in ${@:-2 4 8 16}; do
  line="band_rows $b:"
  for r in 0 1 2 3 4 5 6 7; do
    ms=$(timeout -k 10 300 python bench.py --config C3 --steps 4 --cpu-seconds 0 --no-roofline-pass --simulate-rank $r --simulate-world 8 --band-rows $b 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    line="$line $ms"
  done
  echo "$line"
done
