#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_baseline_configs.py -m gpu -x -q 2>&1 | tail -2
timeout -k 10 200 python scripts/gpu_frame_time.py 2>&1 | grep "megakernel\|auto"
timeout -k 10 100 python scripts/gpu_small_latency.py 2>&1 | grep megakernel
timeout -k 10 300 python bench.py --config C2 --cpu-seconds 0 --no-roofline-pass --steps 3 2>/dev/null | cut -c70-200
