#!/usr/bin/env python3
"""Prints the kernel sequence of the last sample batch in a rocprofv3 kernel_trace.csv (durations in us)."""
import csv, glob, os, sys
pat = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_kernels/*/*kernel_trace.csv"
f = max(glob.glob(pat), key=os.path.getmtime)
rows = [r for r in csv.DictReader(open(f)) if "cgpt::" in r["Kernel_Name"]]
gen = [i for i, r in enumerate(rows) if "generate" in r["Kernel_Name"]]
seq = rows[gen[-1]:] if gen else rows
def name(r): return r["Kernel_Name"].split("(")[0].replace("void ", "").replace("cgpt::", "")
print(" ".join("%s:%.0f" % (name(r)[:10], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in seq))
t0, t1 = int(seq[0]["Start_Timestamp"]), int(seq[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seq)
print("sequence wall %.0f us, sum of kernels %.0f us" % ((t1 - t0) / 1e3, busy / 1e3))
tot = {}
for r in seq:
    tot[name(r)] = tot.get(name(r), 0) + (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print({k: round(v) for k, v in tot.items()})
