#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for cfg in "wavefront CGPT_WF_FUSED=0" "wavefront CGPT_WF_FUSED=1" "wavefront CGPT_WF_FUSED=0" "persistent CGPT_PT_FUSED=0" "persistent CGPT_PT_FUSED=1"; do
  set -- $cfg
  echo "== $cfg"
  env $2 timeout -k 10 300 python bench.py --kernel $1 --cpu-seconds 0 --no-roofline-pass 2> gpurun_out/r02_bench_x.err | cut -c1-200
done
echo "== frame time, fused 0"
timeout -k 10 200 python scripts/gpu_frame_time.py 2>&1 | grep "wavefront\|persistent"
echo "== frame time, fused 1 (drain)"
CGPT_WF_FUSED=1 CGPT_PT_FUSED=1 timeout -k 10 200 python scripts/gpu_frame_time.py 2>&1 | grep "wavefront\|persistent"
