#!/usr/bin/env python3
"""Kernel sequence of the LAST render call in a rocprofv3 kernel_trace.csv: start offset, duration and gap to the previous kernel (us)."""
import csv, glob, os, sys
f = max(glob.glob(sys.argv[1]), key=os.path.getmtime)
rows = [r for r in csv.DictReader(open(f)) if "cgpt::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
acc = [i for i, r in enumerate(rows) if "accumulate" in r["Kernel_Name"] or "megakernel" in r["Kernel_Name"]]
start = acc[-2] + 1 if len(acc) > 1 else 0
seq = rows[start:]
t0 = int(seq[0]["Start_Timestamp"]); prev_end = t0
tot_k = tot_gap = 0.0
for r in seq:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("cgpt::", "")
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:7.1f}  grid {r.get('Grid_Size', '?'):>8}  {name}")
    tot_k += (e - s) / 1e3; tot_gap += max(0, s - prev_end) / 1e3
    prev_end = max(prev_end, e)
print(f"wall {(prev_end - t0) / 1e3:.1f} us, kernels {tot_k:.1f} us, gaps {tot_gap:.1f} us")
