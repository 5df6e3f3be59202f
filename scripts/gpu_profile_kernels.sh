#!/bin/bash
# On the GPU box: rocprofv3 kernel trace of one isolated batch sequence (1 pool, 32 spp).  Extra args go to bench.py.
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_kernels
rm -rf $OUT
cd /tmp && export TMPDIR=/tmp
export CGPT_WF_POOLS=${CGPT_WF_POOLS:-1}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-seconds 0 --spp 32 "$@" > $OUT.log 2>&1
cat $OUT/*/*kernel_stats.csv | cut -c1-200
