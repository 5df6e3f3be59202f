#!/bin/bash
cd $GRAFT_REPO_ROOT
CGPT_WF_EAGER=1 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "wavefront or kernel" 2>&1 | tail -2
run() { echo "== $ENVS $*"; env $ENVS CGPT_WF_PROFILE=1 timeout -k 10 400 python bench.py --cpu-seconds 0 --no-roofline-pass "$@" 2>gpurun_out/eager.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(' ', d['ms_per_step'], 'ms', d['value'], 'Mrays/s')"; grep profile gpurun_out/eager.err; }
for i in 1 2; do
for e in 0 1; do
ENVS="CGPT_WF_EAGER=$e" run --config C3 --steps 3
done
done
ENVS="CGPT_WF_EAGER=1 CGPT_WF_REFILL=32" run --config C3 --steps 3
ENVS="CGPT_WF_EAGER=1 CGPT_WF_INNER_REPEAT=28" run --config C3 --steps 3
ENVS="CGPT_WF_EAGER=1 CGPT_WF_OBJ_REPEAT=8" run --config C3 --steps 3
ENVS="CGPT_WF_EAGER=0" run --config C4 --steps 2 --simulate-rank 2
ENVS="CGPT_WF_EAGER=1" run --config C4 --steps 2 --simulate-rank 2
