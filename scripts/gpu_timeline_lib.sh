#!/bin/bash
# On the GPU box: kernel timeline of the production render (two batches in flight) for one build of the library.
# usage: gpurun -- bash scripts/gpu_timeline_lib.sh <lib.so under cpugpupathtracing_amd/lib> [bench args...]
R=$GRAFT_REPO_ROOT
L=$1; shift
export CGPT_LIB_PATH=$R/cpugpupathtracing_amd/lib/$L
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/timeline_$(basename $L .so)
rm -rf $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --cpu-seconds 0 --no-roofline-pass --steps 2 --warmup 1 "$@" > $OUT.log 2>&1
grep '^{' $OUT.log | cut -c70-160
python3 $R/scripts/analyze_overlap.py "$OUT/*/*kernel_trace.csv" 2 -v > $OUT.txt
head -12 $OUT.txt
