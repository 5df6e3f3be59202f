#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python bench.py --issue-table > gpurun_out/r02_bench_b.json 2> gpurun_out/r02_bench_b.err; echo "bench rc=$?"; cat gpurun_out/r02_bench_b.json; grep "issue rate" gpurun_out/r02_bench_b.err
echo "== scalar slab variant"
CGPT_LIB_PATH=$GRAFT_REPO_ROOT/cpugpupathtracing_amd/lib/libcpugpupt_scalar.so timeout -k 10 300 python bench.py --cpu-seconds 0 --no-roofline-pass 2>/dev/null | cut -c1-330
echo "== packed again"
timeout -k 10 300 python bench.py --cpu-seconds 0 --no-roofline-pass 2>/dev/null | cut -c1-330
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/frame_trace
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/frame_trace -- python3 $R/scripts/gpu_frame_render.py 1 wavefront > $R/gpurun_out/frame_trace.log 2>&1
cd $R && python scripts/frame_timeline.py "gpurun_out/frame_trace/*/*kernel_trace.csv"
