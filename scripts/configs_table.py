#!/usr/bin/env python3
"""gpurun_out/configs.jsonl (scripts/gpu_configs.sh) -> profiles/<round>/bench_configs_C1_C5.jsonl + bench_configs_table.md
usage: python scripts/configs_table.py r02"""
import json, os, re, shutil, sys

rnd = sys.argv[1]
src = "gpurun_out/configs.jsonl"
out = os.path.join("profiles", rnd)
rows = []
for line in open(src):
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    c = d["config"]
    name = c["workload"].split(":")[0]
    m = re.search(r"rank (\d+) of (\d+)", c["workload"])
    share = f"{m.group(1)} of {m.group(2)}" if m else "whole frame"
    rows.append((name, c.get("kernel", ""), share, c.get("rows_per_gpu", ""), d["ms_per_step"], d["value"], c.get("rays_per_step", "")))
shutil.copy(src, os.path.join(out, "bench_configs_C1_C5.jsonl"))
with open(os.path.join(out, "bench_configs_table.md"), "w") as f:
    f.write("# BASELINE.json configurations on one MI355X (`scripts/gpu_configs.sh`; lines in bench_configs_C1_C5.jsonl)\n\n"
            "C4 / C5 are 8-GPU configurations: every rank's interleaved share (4-row bands) is rendered alone on the one GPU of the box, at the\n"
            "stated 1024 / 4096 spp; the frame time of an 8-GPU job is the slowest share plus one 4-16 MB gather.  C1's line carries the CPU\n"
            "oracle beside it (configs[0] is the reference's CPU path).\n\n"
            "| config | kernel | share | rows | ms per step | Mrays/s | rays per step |\n|---|---|---|---|---|---|---|\n")
    for r in rows:
        f.write(f"| {r[0]} | {r[1]} | {r[2]} | {r[3]} | {r[4]:.3f} | {r[5]:.0f} | {r[6]} |\n")
print(len(rows), "rows")
