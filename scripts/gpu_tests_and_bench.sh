#!/bin/bash
# On the GPU box: GPU parity tests, then a short wavefront bench.  usage: gpurun -- bash scripts/gpu_tests_and_bench.sh
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
timeout -k 10 300 python bench.py --steps 2 --warmup 1 --cpu-seconds 0 "$@" | tee gpurun_out/bench_quick.json
