#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_persistent.py -m gpu -x -q 2>&1 | tail -2
timeout -k 10 200 python scripts/gpu_frame_time.py 2>&1 | grep -v amdgpu | tee gpurun_out/final_frame_time.txt | grep "persistent"
timeout -k 10 100 python scripts/gpu_small_latency.py 2>&1 | grep -v amdgpu | tee gpurun_out/final_small_latency.txt | grep persistent
