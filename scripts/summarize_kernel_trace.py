#!/usr/bin/env python3
"""Prints the cgpt kernel sequence of the last render in a rocprofv3 kernel_trace.csv (durations in us) and per-kernel totals."""
import csv, glob, os, sys
pat = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_kernels/*/*kernel_trace.csv"
f = max(glob.glob(pat), key=os.path.getmtime)
rows = [r for r in csv.DictReader(open(f)) if "cgpt::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def name(r): return r["Kernel_Name"].split("(")[0].replace("void ", "").replace("cgpt::", "")
def dur(r): return (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
# the last render = everything after the last-but-one wf_accumulate gap larger than the others; simpler: the last n rows
# holding the final quarter of the accumulates
acc = [i for i, r in enumerate(rows) if "accumulate" in r["Kernel_Name"]]
n_acc = max(1, int(os.environ.get("ACCS_PER_RENDER", "2")))
start = acc[-n_acc - 1] + 1 if len(acc) > n_acc else 0
seq = rows[start:]
print(" ".join("%s:%.0f" % (name(r)[:14], dur(r)) for r in seq))
t0, t1 = int(seq[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in seq)
print("sequence wall %.0f us, sum of kernels %.0f us" % ((t1 - t0) / 1e3, sum(dur(r) for r in seq)))
tot = {}
for r in seq:
    tot[name(r)] = tot.get(name(r), 0) + dur(r)
print({k: round(v) for k, v in tot.items()})
