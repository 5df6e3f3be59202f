#!/bin/bash
# On the GPU box: per-kernel durations (rocprofv3 --kernel-trace --stats, one batch in flight) of the default bench under several
# environment settings.  usage: gpurun -- bash scripts/gpu_kernel_stats_envs.sh "CGPT_WF_BANDS=1" "CGPT_WF_BANDS=8" ...
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for E in "$@"; do
  OUT=$R/gpurun_out/envstats_$(echo $E | tr -c 'A-Za-z0-9_\n' '_')
  rm -rf $OUT
  export $E
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --cpu-seconds 0 --no-roofline-pass --steps 1 --warmup 1 --pools 1 > $OUT.log 2>&1
  echo "== $E"; grep '^{' $OUT.log | cut -c70-160
  python3 - "$OUT" <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    n=r['Name'].split('(')[0].replace('void ','').replace('cgpt::','')
    if 'true' in n.split('<')[-1].split(',')[0] or 'rocclr' in n: continue
    print(f"  {n:28s} calls {r['Calls']:>4s} total {float(r['TotalDurationNs'])/1e6:9.3f} ms  avg {float(r['AverageNs'])/1e3:10.1f} us")
PY
done
