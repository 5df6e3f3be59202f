#!/bin/bash
# On the GPU box: PMC passes (SQ activity, instruction mix, L1/L2) for one isolated wavefront batch sequence (1 pool, 16 spp).
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export CGPT_WF_POOLS=1
B="python3 $R/bench.py --steps 1 --warmup 1 --cpu-seconds 0 --spp ${SPP:-16} --kernel wavefront"
rm -rf $R/gpurun_out/pmc_a $R/gpurun_out/pmc_b $R/gpurun_out/pmc_c
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pmc_a -- $B > $R/gpurun_out/pmc_a.log 2>&1
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/pmc_b -- $B > $R/gpurun_out/pmc_b.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --output-format csv -d $R/gpurun_out/pmc_c -- $B > $R/gpurun_out/pmc_c.log 2>&1
echo ok
