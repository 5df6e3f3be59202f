#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_baseline_configs.py -m gpu -x -q -k "wavefront or kernel or c3 or c2" 2>&1 | tail -2
run() { echo "== $ENVS $*"; env $ENVS timeout -k 10 400 python bench.py --cpu-seconds 0 --no-roofline-pass "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(' ', d['ms_per_step'], 'ms', d['value'], 'Mrays/s')"; }
for i in 1 2; do
for e in 0 1; do
ENVS="CGPT_WF_RETIRE_MISSES=$e" run --config C3 --steps 3
ENVS="CGPT_WF_RETIRE_MISSES=$e" run --config C3 --steps 3 --pools 1
done
done
ENVS="CGPT_WF_RETIRE_MISSES=0" run --config C4 --steps 2 --simulate-rank 2
ENVS="CGPT_WF_RETIRE_MISSES=1" run --config C4 --steps 2 --simulate-rank 2
