#!/bin/bash
# On the GPU box: run selected GPU tests with timing.  usage: gpurun -- bash scripts/gpu_pytest.sh tests/test_baseline_configs.py
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest "$@" -m gpu -x -q --durations=8 2>&1 | tail -30
