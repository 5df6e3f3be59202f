#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_persistent.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
run() { echo "== $ENVS $*"; env $ENVS timeout -k 10 400 python bench.py --cpu-seconds 0 --no-roofline-pass "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(' ', d['ms_per_step'], 'ms', d['value'], 'Mrays/s')"; }
for i in 1 2; do
for dyn in 0 1; do
ENVS="CGPT_WF_DYNAMIC=$dyn" run --config C3 --steps 3
ENVS="CGPT_WF_DYNAMIC=$dyn" run --config C3 --steps 3 --simulate-rank 2 --simulate-world 8
ENVS="CGPT_WF_DYNAMIC=$dyn" run --config C4 --steps 2 --simulate-rank 2
done
done
echo "== frame time dynamic 0"; CGPT_WF_DYNAMIC=0 timeout -k 10 200 python scripts/gpu_frame_time.py 2>&1 | grep "wavefront"
echo "== frame time dynamic 1"; CGPT_WF_DYNAMIC=1 timeout -k 10 200 python scripts/gpu_frame_time.py 2>&1 | grep "wavefront"
