#!/bin/bash
# On the GPU box: the whole GPU suite, the default bench line, the persistent-kernel and COMPARISON-mode lines, frame times of small calls
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout -k 10 500 python bench.py > gpurun_out/final_bench_default.json 2> gpurun_out/final_bench_default.err; echo "bench rc=$?"; cat gpurun_out/final_bench_default.json
timeout -k 10 300 python bench.py --kernel persistent --cpu-seconds 0 > gpurun_out/final_bench_persistent.json 2>/dev/null; cut -c1-200 gpurun_out/final_bench_persistent.json
timeout -k 10 300 python bench.py --mode comparison --kernel auto --cpu-seconds 0 --no-roofline-pass > gpurun_out/final_bench_comparison.json 2>/dev/null; cut -c1-200 gpurun_out/final_bench_comparison.json
timeout -k 10 200 python scripts/gpu_frame_time.py 2>&1 | grep -v amdgpu > gpurun_out/final_frame_time.txt; cat gpurun_out/final_frame_time.txt
timeout -k 10 100 python scripts/gpu_small_latency.py 2>&1 | grep -v amdgpu > gpurun_out/final_small_latency.txt
timeout -k 10 100 python __graft_entry__.py smoke 2>&1 | tail -2
