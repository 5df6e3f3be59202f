#!/bin/bash
# On the GPU box: kernel trace of the default bench workload (all pools), timeline summary of the last render.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_full
rm -rf $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 "$@" > $OUT.log 2>&1
cd $R && python scripts/analyze_timeline.py "gpurun_out/prof_full/*/*kernel_trace.csv" ${ACCS:-2}
