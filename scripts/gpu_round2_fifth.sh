#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_persistent.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r02_pytest_pt.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r02_pytest_pt.log
timeout -k 10 200 python scripts/gpu_frame_time.py > gpurun_out/r02_frame_time3.txt 2>&1; grep -v megakernel gpurun_out/r02_frame_time3.txt
echo "== fused everywhere"
CGPT_WF_FUSED=1 CGPT_PT_FUSED=1 timeout -k 10 200 python scripts/gpu_frame_time.py 2>&1 | grep "wavefront\|persistent"
for cfg in "wavefront" "wavefront CGPT_WF_FUSED=1" "persistent" "persistent CGPT_PT_FUSED=1"; do
  set -- $cfg
  echo "== $cfg"
  env $2 CGPT_WF_PROFILE=1 timeout -k 10 300 python bench.py --kernel $1 --cpu-seconds 0 --no-roofline-pass 2> gpurun_out/r02_bench_x.err | cut -c1-200; grep profile gpurun_out/r02_bench_x.err
done
