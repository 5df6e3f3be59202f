#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python scripts/gpu_issue_table.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_issue_table.txt
bash scripts/gpu_roofline_pmc.sh c3 --config C3
