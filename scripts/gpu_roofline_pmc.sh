#!/bin/bash
# On the GPU box: the inputs of bench.py's roofline for one workload, all from the SAME command line with one batch in flight
# (--pools 1: dispatches do not overlap, so per-kernel counters and durations are chip-exclusive):
#   1. rocprofv3 --kernel-trace --stats      -> per-kernel durations (must agree with bench.py's hipEvents)
#   2. rocprofv3 --pmc SQ_* (one pass)       -> VALU wave-instructions, active lanes, wave cycles
#   3. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, MI355X_MICROARCH.md "HBM")
#   4. rocprofv3 --pmc GRBM_GUI_ACTIVE TD_TD_BUSY_sum TD_TC_STALL_sum -> how busy the vector-memory return path is (mem_return_frac)
# then scripts/summarize_roofline_pmc.py folds them into profiles/pmc_counts.json + profiles/<round>/.
# usage: gpurun -- [INSTALL=r03] bash scripts/gpu_roofline_pmc.sh <tag> [bench.py args...]     e.g.  c3   or   c4 --config C4
# INSTALL=<round>: also copy the summary to profiles/<round>/ and merge the entry (with the hash of the device sources) into profiles/pmc_counts.json
R=$GRAFT_REPO_ROOT
TAG=${1:-c3}; shift
ARGS="--steps 1 --warmup 1 --cpu-seconds 0 --no-roofline-pass --pools 1 $*"
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for pass in kernel sq fetch write td; do
  OUT=$R/gpurun_out/roof_${TAG}_$pass
  rm -rf $OUT
  case $pass in
    kernel) OPTS="--kernel-trace --stats" ;;
    sq)     OPTS="--pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM" ;;
    fetch)  OPTS="--pmc FETCH_SIZE" ;;
    write)  OPTS="--pmc WRITE_SIZE" ;;
    td)     OPTS="--pmc GRBM_GUI_ACTIVE TD_TD_BUSY_sum TD_TC_STALL_sum" ;;
  esac
  echo "== pass $pass"
  timeout -k 10 ${PASS_TIMEOUT:-300} rocprofv3 $OPTS --output-format csv -d $OUT -- python3 $R/bench.py $ARGS > $OUT.log 2>&1 || { echo "pass $pass failed"; tail -5 $OUT.log; exit 1; }
  grep '^{' $OUT.log | tail -1 | cut -c1-400
done
cd $R && python3 scripts/summarize_roofline_pmc.py $TAG gpurun_out/roofline_$TAG ${INSTALL:+--install $INSTALL}
