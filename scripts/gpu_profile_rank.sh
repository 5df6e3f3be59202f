#!/bin/bash
# On the GPU box: kernel trace of one rank's share of an 8-GPU job (default rank 0), all pools.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_rank
rm -rf $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --simulate-rank ${1:-0} --simulate-world 8 > $OUT.log 2>&1
cd $R && python scripts/analyze_timeline.py "gpurun_out/prof_rank/*/*kernel_trace.csv" 4
