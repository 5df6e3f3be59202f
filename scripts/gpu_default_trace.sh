#!/bin/bash
# On the GPU box: rocprofv3 --kernel-trace --stats of the driver's own bench command (python bench.py: 1 warm-up + 3 timed steps with
# two batches in flight, then bench.py's untimed single-pool roofline pass and the issue-rate microbenchmark).
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/default_trace
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/default_trace -- python3 $R/bench.py --cpu-seconds 0 > $R/gpurun_out/default_trace.json 2> $R/gpurun_out/default_trace.err
cp $R/gpurun_out/default_trace/*/*kernel_stats.csv $R/gpurun_out/default_trace_kernel_stats.csv
cut -c1-160 $R/gpurun_out/default_trace_kernel_stats.csv | head -14
cut -c1-400 $R/gpurun_out/default_trace.json
