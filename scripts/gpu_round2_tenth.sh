#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_persistent.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
timeout -k 10 200 python scripts/gpu_frame_time.py 2>&1 | grep "persistent\|wavefront"
for fr in 0 8 32; do echo "fine_rounds $fr"; timeout -k 10 200 python scripts/gpu_frame_time.py 1920 1080 pt_fine_rounds=$fr 2>&1 | grep "persistent"; done
echo "== persistent 256 spp"
timeout -k 10 300 python bench.py --cpu-seconds 0 --no-roofline-pass --steps 3 --kernel persistent 2>/dev/null | cut -c70-200
