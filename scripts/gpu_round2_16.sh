#!/bin/bash
cd $GRAFT_REPO_ROOT
CGPT_LIB_PATH=$GRAFT_REPO_ROOT/cpugpupathtracing_amd/lib/libcpugpupt_soa.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "matches_oracle or intersect_rays" 2>&1 | tail -2
bash scripts/gpu_ab_libs.sh cpugpupathtracing_amd/lib/libcpugpupt.so cpugpupathtracing_amd/lib/libcpugpupt_soa.so --config C3
bash scripts/gpu_ab_libs.sh cpugpupathtracing_amd/lib/libcpugpupt.so cpugpupathtracing_amd/lib/libcpugpupt_soa.so --config C4 --simulate-rank 2
