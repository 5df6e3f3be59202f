#!/usr/bin/env python3
"""How well the two batches of a production render overlap: for the LAST render in a rocprofv3 kernel_trace.csv, the wall time, the time with
at least one wf_trace running, and what ran while none was.  usage: analyze_overlap.py <glob of kernel_trace.csv> [accumulates per render=4] [-v]"""
import csv, glob, os, sys
f = max(glob.glob(sys.argv[1]), key=os.path.getmtime)
n_acc = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 4
rows = [r for r in csv.DictReader(open(f)) if "cgpt::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def name(r): return r["Kernel_Name"].split("(")[0].replace("void ", "").replace("cgpt::", "").replace(" ", "")
ends = sorted(int(r["End_Timestamp"]) for r in rows if "accumulate" in r["Kernel_Name"])
t_prev = ends[-n_acc - 1] if len(ends) > n_acc else 0
seq = [r for r in rows if int(r["Start_Timestamp"]) >= t_prev]
t0 = min(int(r["Start_Timestamp"]) for r in seq); t1 = max(int(r["End_Timestamp"]) for r in seq)
def cover(sel):
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seq if sel(r))
    tot = 0; cur_s = cur_e = None; merged = []
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None: merged.append((cur_s, cur_e))
            cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    if cur_e is not None: merged.append((cur_s, cur_e))
    return sum(e - s for s, e in merged) / 1e6, merged
tr, tr_iv = cover(lambda r: "wf_trace" in r["Kernel_Name"])
any_, _ = cover(lambda r: True)
print(f"render wall {(t1 - t0) / 1e6:.2f} ms | >=1 kernel {any_:.2f} | >=1 wf_trace {tr:.2f} | no trace running {(t1 - t0) / 1e6 - tr:.2f} ms")
print("sum of durations:", {n: round(sum((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) for r in seq if name(r) == n) / 1e6, 2) for n in sorted({name(r) for r in seq})})
# gaps between trace intervals and what ran inside them
gaps = []
prev = t0
for s, e in tr_iv:
    if s > prev: gaps.append((prev, s))
    prev = e
if t1 > prev: gaps.append((prev, t1))
big = sorted(gaps, key=lambda g: g[0] - g[1])[:12]
for s, e in sorted(big):
    inside = {}
    for r in seq:
        a, b = max(s, int(r["Start_Timestamp"])), min(e, int(r["End_Timestamp"]))
        if b > a: inside[name(r)] = inside.get(name(r), 0) + (b - a) / 1e3
    print(f"  gap at {(s - t0) / 1e6:7.2f} ms, {(e - s) / 1e3:7.1f} us:", {k: round(v) for k, v in inside.items()})
if "-v" in sys.argv:
    for r in seq:
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {(int(r['End_Timestamp']) - t0) / 1e3:9.1f} q{r.get('Queue_Id', '?')} {name(r)}")
