#!/bin/bash
# On the GPU box: kernel timeline of the last of six n-sample render calls.  usage: gpurun -- bash scripts/gpu_frame_trace.sh <n> <kernel>
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/frame_trace
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/frame_trace -- python3 $R/scripts/gpu_frame_render.py ${1:-1} ${2:-wavefront} > $R/gpurun_out/frame_trace.log 2>&1
cd $R && python scripts/frame_timeline.py "gpurun_out/frame_trace/*/*kernel_trace.csv"
