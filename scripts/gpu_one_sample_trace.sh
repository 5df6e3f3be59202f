#!/bin/bash
# On the GPU box: kernel trace of one-sample 1080p calls per kernel choice (which launches a call is made of, and how long each takes).
# usage: gpurun -- bash scripts/gpu_one_sample_trace.sh [knob=value ...]
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/one_sample_trace
rm -rf $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/scripts/gpu_frame_time.py samples=1 "$@" > $OUT.log 2>&1
grep "n_samples" $OUT.log
python3 - $OUT <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
agg = collections.OrderedDict()
for r in rows:
    n = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('cgpt::', '')
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    a = agg.setdefault(n, [0, 0.0, 1e9, 0.0]); a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
for n, (c, t, lo, hi) in agg.items():
    print(f"{n:40s} calls {c:4d} avg {t / c:9.1f} us  min {lo:9.1f}  max {hi:9.1f}")
PY
