#!/usr/bin/env python3
"""Copies the summaries of a scripts/gpu_full_bench.sh run (gpurun_out/*_<tag>*) into profiles/<round>/ and refreshes
profiles/hbm_traffic.json.  usage: python scripts/save_profiles.py <tag> [round_dir]"""
import collections, csv, glob, json, os, shutil, sys
tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r01"
out = os.path.join("profiles", rnd)
os.makedirs(out, exist_ok=True)
ks = max(glob.glob(f"gpurun_out/prof_{tag}_kernel/*/*kernel_stats.csv"), key=os.path.getmtime)
shutil.copy(ks, os.path.join(out, "wavefront_kernel_stats.csv"))
shutil.copy(f"gpurun_out/bench_{tag}.json", os.path.join(out, "wavefront_bench.json"))

def load(kind):
    f = max(glob.glob(f"gpurun_out/prof_{tag}_{kind}/*/*counter_collection.csv"), key=os.path.getmtime)
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("cgpt::", "")
        if "wf_" in k or "megakernel" in k:
            agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
    return agg

fe, wr = load("fetch"), load("write")
with open(os.path.join(out, "wavefront_pmc_summary.csv"), "w") as f:
    f.write("kernel,calls,FETCH_SIZE_KiB_total,WRITE_SIZE_KiB_total,hbm_bytes_per_call=(2*FETCH+WRITE)*1024/calls\n")
    for k in sorted(fe):
        f.write(f"{k},{fe[k][1]},{fe[k][0]:.0f},{wr[k][0]:.0f},{(2 * fe[k][0] + wr[k][0]) * 1024 / fe[k][1]:.0f}\n")
sel = [k for k in fe if k.startswith("wf_trace<false")]
n = sum(fe[k][1] for k in sel)
per = (2 * sum(fe[k][0] for k in sel) + sum(wr[k][0] for k in sel)) * 1024 / n
whole = sum((2 * fe[k][0] + wr[k][0]) * 1024 for k in fe if "<true" not in k)
j = json.load(open("profiles/hbm_traffic.json"))
j["1920x1080x256_l6_m3_wf_trace_n1"] = int(per)
json.dump(j, open("profiles/hbm_traffic.json", "w"), indent=2)
print(f"wf_trace: {per / 1e9:.3f} GB of fabric traffic per launch over {n} launches; whole render {whole / 1e9:.1f} GB")
print(open(os.path.join(out, "wavefront_pmc_summary.csv")).read())
