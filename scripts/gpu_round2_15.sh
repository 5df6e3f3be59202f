#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "octant or wavefront" 2>&1 | tail -3
run() { echo "== $ENVS $*"; env $ENVS CGPT_WF_PROFILE=1 timeout -k 10 400 python bench.py --cpu-seconds 0 --no-roofline-pass "$@" 2>gpurun_out/sort.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms', d['value'], 'Mrays/s')"; grep profile gpurun_out/sort.err; }
for i in 1 2; do
ENVS="CGPT_WF_SORT=0" run --config C4 --steps 2 --simulate-rank 2
ENVS="CGPT_WF_SORT=1" run --config C4 --steps 2 --simulate-rank 2
done
ENVS="CGPT_WF_SORT=0" run --config C3 --steps 3
ENVS="CGPT_WF_SORT=1" run --config C3 --steps 3
ENVS="CGPT_WF_SORT=0" run --config C5 --steps 1 --simulate-rank 2
ENVS="CGPT_WF_SORT=1" run --config C5 --steps 1 --simulate-rank 2
