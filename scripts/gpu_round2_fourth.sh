#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_persistent.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r02_pytest_pt.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/r02_pytest_pt.log
timeout -k 10 200 python scripts/gpu_frame_time.py > gpurun_out/r02_frame_time2.txt 2>&1; cat gpurun_out/r02_frame_time2.txt
for k in wavefront persistent; do
  CGPT_WF_PROFILE=1 timeout -k 10 300 python bench.py --kernel $k --cpu-seconds 0 --no-roofline-pass 2> gpurun_out/r02_bench_$k.err | cut -c1-330; grep profile gpurun_out/r02_bench_$k.err
done
