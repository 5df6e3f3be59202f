#!/usr/bin/env python3
"""Timeline of the cgpt kernels of the LAST render in a rocprofv3 kernel_trace.csv: wall time, time with >= 1 kernel running,
sum of durations per kernel name, mean concurrency.  usage: analyze_timeline.py <glob of kernel_trace.csv> <accumulates per render>"""
import csv, glob, os, sys
f = max(glob.glob(sys.argv[1]), key=os.path.getmtime)
n_acc = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows = [r for r in csv.DictReader(open(f)) if "cgpt::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
acc = [i for i, r in enumerate(rows) if "accumulate" in r["Kernel_Name"]]
ends = sorted(int(rows[i]["End_Timestamp"]) for i in acc)
t_prev_end = ends[-n_acc - 1] if len(ends) > n_acc else 0
seq = [r for r in rows if int(r["Start_Timestamp"]) >= t_prev_end]
def name(r): return r["Kernel_Name"].split("(")[0].replace("void ", "").replace("cgpt::", "")
ev = []
for r in seq:
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
t0, t1 = ev[0][0], ev[-1][0]
busy = 0; conc_area = 0; depth = 0; last = t0
for t, d in ev:
    if depth > 0: busy += t - last
    conc_area += depth * (t - last)
    depth += d; last = t
tot = {}
for r in seq:
    tot[name(r)] = tot.get(name(r), 0) + (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("kernels %d, wall %.0f us, >=1 kernel running %.0f us (idle %.0f us), mean concurrency while busy %.2f" % (len(seq), (t1 - t0) / 1e3, busy / 1e3, (t1 - t0 - busy) / 1e3, conc_area / max(1, busy)))
print({k: round(v) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])})
