#!/bin/bash
# On the GPU box: the tables behind DESIGN.md section 5 / 6 at the end of a round: calls of 1..128 samples per kernel, small calls over four
# frame sizes, every BASELINE configuration and rank share, wave-time and memory-path counters, the production kernel timeline.
# usage: gpurun --timeout 1190 -- bash scripts/gpu_final_tables.sh
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 200 python scripts/gpu_frame_time.py samples=1,2,4,8,16,32,64,128 2>&1 | grep -v amdgpu > gpurun_out/final_frame_time.txt; tail -3 gpurun_out/final_frame_time.txt
timeout -k 10 200 python scripts/gpu_small_calls.py 2>&1 | grep -v amdgpu > gpurun_out/final_small_calls.txt; tail -2 gpurun_out/final_small_calls.txt
timeout -k 10 200 python bench.py --mode brute --cpu-seconds 0 --no-roofline-pass > gpurun_out/final_bench_brute.json 2>/dev/null; cut -c60-170 gpurun_out/final_bench_brute.json
timeout -k 10 200 python bench.py --mode comparison --kernel persistent --cpu-seconds 0 --no-roofline-pass > gpurun_out/final_bench_comparison_pt.json 2>/dev/null; cut -c60-170 gpurun_out/final_bench_comparison_pt.json
bash scripts/gpu_wave_time_pmc.sh c3 --config C3 > gpurun_out/final_wave_time.txt 2>&1; bash scripts/gpu_wave_time_pmc.sh c3pt --config C3 --kernel persistent | grep pt_persistent >> gpurun_out/final_wave_time.txt; cat gpurun_out/final_wave_time.txt
bash scripts/gpu_mem_path_pmc.sh c3 --config C3 > gpurun_out/final_mem_path.txt 2>&1; cat gpurun_out/final_mem_path.txt
bash scripts/gpu_timeline_lib.sh libcpugpupt.so > /dev/null 2>&1; head -8 gpurun_out/timeline_libcpugpupt.txt
bash scripts/gpu_configs.sh 2>&1 | tail -40
