#!/bin/bash
# On the GPU box: where the waves of each kernel spend their cycles (SQ counters, one batch in flight): parked at s_waitcnt / barriers
# (SQ_WAIT_ANY), stalled at issue (SQ_WAIT_INST_ANY, of which LDS: SQ_WAIT_INST_LDS), issuing (SQ_ACTIVE_INST_ANY); LDS bank conflicts.
# usage: gpurun -- bash scripts/gpu_wave_time_pmc.sh <tag> [bench.py args...]
R=$GRAFT_REPO_ROOT
TAG=${1:-c3}; shift
ARGS="--steps 1 --warmup 1 --cpu-seconds 0 --no-roofline-pass --pools 1 $*"
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/wavetime_$TAG
rm -rf $OUT
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT -- python3 $R/bench.py $ARGS > $OUT.log 2>&1 || { echo failed; tail -5 $OUT.log; exit 1; }
python3 - "$OUT" <<'PY'
import csv,glob,collections,sys
f=max(glob.glob(sys.argv[1]+'/*/*counter_collection.csv'))
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].split('(')[0].replace('void ','').replace('cgpt::','').replace(' ','')
    if '<true' in k or 'rocclr' in k: continue
    agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
print(f"{'kernel':26s} {'wave cycles':>14s}  parked  issue-stall (LDS)  issuing  VALU-issuing | LDS conflict / LDS active")
for k,v in sorted(agg.items(), key=lambda kv:-kv[1]['SQ_WAVE_CYCLES']):
    w=v['SQ_WAVE_CYCLES'] or 1
    print(f"{k:26s} {w:14.3e}  {v['SQ_WAIT_ANY']/w:6.3f}  {v['SQ_WAIT_INST_ANY']/w:6.3f} ({v['SQ_WAIT_INST_LDS']/w:5.3f})  {v['SQ_ACTIVE_INST_ANY']/w:7.3f}  {v['SQ_ACTIVE_INST_VALU']/w:7.3f}      | {v['SQ_LDS_BANK_CONFLICT']:.3e} / {v['SQ_LDS_IDX_ACTIVE']:.3e} = {v['SQ_LDS_BANK_CONFLICT']/max(v['SQ_LDS_IDX_ACTIVE'],1):.3f}")
PY
