set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/counters_list.txt 2>&1 || true
B="python3 $R/bench.py --steps 1 --warmup 1 --cpu-seconds 0 --spp 32"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pmc_a -- $B > $R/gpurun_out/pmc_a.log 2>&1
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $R/gpurun_out/pmc_b -- $B > $R/gpurun_out/pmc_b.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $R/gpurun_out/pmc_c -- $B > $R/gpurun_out/pmc_c.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $R/gpurun_out/pmc_d -- $B > $R/gpurun_out/pmc_d.log 2>&1
echo done
