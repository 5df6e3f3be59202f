#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_gpu_persistent.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
CGPT_WF_TILE_MAJOR=1 timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "wavefront or kernel" 2>&1 | tail -2
run() { echo "== $ENVS $*"; env $ENVS timeout -k 10 400 python bench.py --cpu-seconds 0 --no-roofline-pass "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(' ', d['ms_per_step'], 'ms', d['value'], 'Mrays/s')"; }
for i in 1 2; do
for t in 0 1; do
ENVS="CGPT_WF_TILE_MAJOR=$t" run --config C3 --steps 3
ENVS="CGPT_WF_TILE_MAJOR=$t" run --config C4 --steps 2 --simulate-rank 2
ENVS="CGPT_WF_TILE_MAJOR=$t" run --config C3 --steps 4 --simulate-rank 2 --simulate-world 8
done
done
ENVS="A=1" run --kernel persistent --steps 3
ENVS="A=1" run --config C4 --kernel persistent --steps 2 --simulate-rank 2
timeout -k 10 200 python scripts/gpu_frame_time.py 2>&1 | grep "persistent"
