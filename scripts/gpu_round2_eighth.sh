#!/bin/bash
cd $GRAFT_REPO_ROOT
bash scripts/gpu_ab_libs.sh cpugpupathtracing_amd/lib/libcpugpupt_old.so cpugpupathtracing_amd/lib/libcpugpupt.so --kernel wavefront
echo "== persistent"
timeout -k 10 300 python bench.py --cpu-seconds 0 --no-roofline-pass --steps 3 --kernel persistent 2>/dev/null | cut -c70-200
timeout -k 10 200 python scripts/gpu_frame_time.py 2>&1 | grep "wavefront\|persistent"
timeout -k 10 600 python -m pytest tests/test_gpu_persistent.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
