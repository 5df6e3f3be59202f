#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for cfg in "wavefront A=1" "persistent A=1" "persistent CGPT_PT_TOP_RECORDS=255" "persistent CGPT_PT_SHADE_SHIFT=1" "persistent CGPT_PT_SHADE_SHIFT=2" "persistent CGPT_PT_REFILL=8" "persistent CGPT_PT_REFILL=32" "persistent CGPT_PT_INNER_REPEAT=32" "persistent CGPT_PT_INNER_REPEAT=12"; do
  set -- $cfg
  echo "== $cfg"
  env $2 timeout -k 10 300 python bench.py --kernel $1 --cpu-seconds 0 --no-roofline-pass --steps 2 2> gpurun_out/r02_bench_x.err | cut -c80-200
done
timeout -k 10 200 python scripts/gpu_frame_time.py 2>&1 | grep "wavefront\|persistent"
