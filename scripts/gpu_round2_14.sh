#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env $ENVS timeout -k 10 400 python bench.py --cpu-seconds 0 --no-roofline-pass "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms', d['value'], 'Mrays/s')"; }
for mb in 128 256 512 1024; do for pools in 8 4 2; do ENVS="CGPT_WF_MAX_BATCH=$mb CGPT_WF_POOLS=$pools" run --config C4 --steps 2 --simulate-rank 2; done; done
for mb in 128 256; do ENVS="CGPT_WF_MAX_BATCH=$mb" run --config C3 --steps 3 --simulate-rank 2 --simulate-world 8; done
ENVS="A=1" run --config C2 --kernel wavefront --steps 5
ENVS="A=1" run --config C2 --kernel persistent --steps 5
ENVS="A=1" run --config C4 --kernel persistent --steps 1 --simulate-rank 2
