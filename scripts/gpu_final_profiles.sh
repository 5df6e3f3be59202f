#!/bin/bash
# On the GPU box: the round's roofline inputs (single-pool PMC + kernel-trace passes) for the default workload with the wavefront
# pipeline and with the persistent kernel, then the default bench line.
cd $GRAFT_REPO_ROOT
bash scripts/gpu_roofline_pmc.sh c3 --config C3 | tail -12
bash scripts/gpu_roofline_pmc.sh c3pt --config C3 --kernel persistent | tail -8
