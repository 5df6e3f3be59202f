#!/bin/bash
# On the GPU box: memory-pipeline PMC passes (TA / TCP / TD busy and stalls, VMEM latency) for one isolated wavefront batch
# sequence.  Two counters per hardware block and pass (more fails with "exceeds the capabilities of the hardware").
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export CGPT_WF_POOLS=1
B="python3 $R/bench.py --steps 1 --warmup 1 --cpu-seconds 0 --spp 16 --kernel wavefront"
i=0
for set in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum" \
           "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmcm_$i
  echo "pass $i: $set"
  timeout -k 5 100 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmcm_$i -- $B > $R/gpurun_out/pmcm_$i.log 2>&1 || { echo "pass $i failed"; grep -i "error code" $R/gpurun_out/pmcm_$i.log | head -2; }
done
echo ok
