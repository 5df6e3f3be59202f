#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_persistent.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do
for cfg in "wavefront CGPT_NO_TINY_MESH=1" "wavefront A=1" "persistent CGPT_NO_TINY_MESH=1" "persistent A=1"; do
  set -- $cfg
  echo "== $cfg"
  env $2 CGPT_WF_PROFILE=1 timeout -k 10 300 python bench.py --kernel $1 --cpu-seconds 0 --no-roofline-pass --steps 3 2> gpurun_out/r02_bench_x.err | cut -c70-200; grep profile gpurun_out/r02_bench_x.err
done
done
