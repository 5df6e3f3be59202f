#!/bin/bash
# On the GPU box: the vector-memory path of every kernel (one batch in flight): TA / TD busy, L1 (TCP) and L2 (TCC) hit rates, average
# L1 and L1->L2 latencies.  Few counters per hardware block and pass.  usage: gpurun -- bash scripts/gpu_mem_path_pmc.sh <tag> [bench args]
R=$GRAFT_REPO_ROOT
TAG=${1:-c3}; shift
ARGS="--steps 1 --warmup 1 --cpu-seconds 0 --no-roofline-pass --pools 1 $*"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum" "TD_TD_BUSY_sum TD_TC_STALL_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/mempath_${TAG}_$i
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/mempath_${TAG}_$i -- python3 $R/bench.py $ARGS > $R/gpurun_out/mempath_${TAG}_$i.log 2>&1 || { echo "pass $i ($set) failed"; grep -i "error" $R/gpurun_out/mempath_${TAG}_$i.log | head -2; }
done
python3 - "$R/gpurun_out/mempath_${TAG}_" <<'PY'
import csv,glob,collections,sys
agg=collections.defaultdict(lambda: collections.defaultdict(float)); calls=collections.Counter()
for i in range(1,6):
    fs=glob.glob(sys.argv[1]+str(i)+'/*/*counter_collection.csv')
    if not fs: continue
    seen=set()
    for r in csv.DictReader(open(max(fs))):
        k=r['Kernel_Name'].split('(')[0].replace('void ','').replace('cgpt::','').replace(' ','')
        if '<true' in k or 'rocclr' in k: continue
        agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
for k,c in sorted(agg.items(), key=lambda kv:-kv[1].get('GRBM_GUI_ACTIVE',0)):
    g=c.get('GRBM_GUI_ACTIVE',0)/8 or 1            # summed over 8 XCDs
    print(f"{k:24s} TA busy {c.get('TA_TA_BUSY_sum',0)/256/g:5.2f}  TD busy {c.get('TD_TD_BUSY_sum',0)/256/g:5.2f} (stalled by TC {c.get('TD_TC_STALL_sum',0)/256/g:5.2f})  "
          f"L1 hit {1-c.get('TCP_TCC_READ_REQ_sum',0)/max(c.get('TCP_TOTAL_CACHE_ACCESSES_sum',1),1):5.2f}  L2 hit {c.get('TCC_HIT_sum',0)/max(c.get('TCC_HIT_sum',0)+c.get('TCC_MISS_sum',0),1):5.2f}  "
          f"L1 latency {c.get('TCP_TCP_LATENCY_sum',0)/max(c.get('TCP_TOTAL_ACCESSES_sum',1),1):7.0f} clk  L1->L2 read latency {c.get('TCP_TCC_READ_REQ_LATENCY_sum',0)/max(c.get('TCP_TCC_READ_REQ_sum',1),1):7.0f} clk")
PY
