#!/usr/bin/env python3
"""Folds the four passes of scripts/gpu_roofline_pmc.sh into one summary directory and the entry bench.py reads.

usage: python scripts/summarize_roofline_pmc.py <tag> <out_dir> [--install <round>]
  reads  gpurun_out/roof_<tag>_{kernel,sq,fetch,write}
  writes <out_dir>/{kernel_stats.csv, pmc_per_kernel.csv, entry.json, bench_line.json}
  --install r02: copies them to profiles/r02/roofline_<tag>_* and merges entry.json into profiles/pmc_counts.json
HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md, HBM / rocprofv3 section: FETCH_SIZE is in KiB and reads
half of the bytes of 16-B-per-lane loads on gfx950)."""
import collections, csv, glob, json, os, shutil, sys

tag, out = sys.argv[1], sys.argv[2]
os.makedirs(out, exist_ok=True)


def short(name):
    return name.split("(")[0].replace("void ", "").replace("cgpt::", "").replace(" ", "")


def newest(pattern):
    fs = glob.glob(pattern)
    if not fs:
        raise SystemExit(f"nothing matches {pattern}")
    return max(fs, key=os.path.getmtime)


def counters(kind):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(newest(f"gpurun_out/roof_{tag}_{kind}/*/*counter_collection.csv"))):
        k = short(r["Kernel_Name"])
        if not ("wf_" in k or "pt_" in k or "megakernel" in k) or "<true" in k:      # the COUNT variants belong to the warm-up step
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        d = (k, r["Dispatch_Id"])
        if d not in seen:
            seen.add(d); calls[k] += 1
    return agg, calls


sq, calls = counters("sq")
fe, _ = counters("fetch")
wr, _ = counters("write")
try:
    td, _ = counters("td")
except SystemExit:                     # an older capture without the TD pass
    td = {}
line = None
for l in open(f"gpurun_out/roof_{tag}_kernel.log"):
    if l.startswith("{"):
        line = json.loads(l)
assert line, "no bench line in the kernel pass log"
json.dump(line, open(os.path.join(out, "bench_line.json"), "w"), indent=1)
shutil.copy(newest(f"gpurun_out/roof_{tag}_kernel/*/*kernel_stats.csv"), os.path.join(out, "kernel_stats.csv"))

# durations of the timed step from the kernel trace
dur = collections.defaultdict(float)
ncall = collections.Counter()
for r in csv.DictReader(open(newest(f"gpurun_out/roof_{tag}_kernel/*/*kernel_trace.csv"))):
    k = short(r["Kernel_Name"])
    if ("wf_" in k or "pt_" in k or "megakernel" in k) and "<true" not in k:
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        ncall[k] += 1

# Kernels without a COUNT variant (accumulate, plan, gather) ran in the warm-up step too (--steps 1 --warmup 1: two identical
# steps, the first with the COUNT instantiations of the templated kernels): halve what they collected so every row is ONE step.
for table in (sq, fe, wr):
    for k in table:
        if "<" not in k:
            for n in table[k]:
                table[k][n] /= 2.0
for k in list(calls):
    if "<" not in k:
        calls[k] //= 2
for k in list(dur):
    if "<" not in k:
        dur[k] /= 2.0

rows = []
for k in sorted(sq):
    c = sq[k]
    hbm = (2 * fe[k].get("FETCH_SIZE", 0.0) + wr[k].get("WRITE_SIZE", 0.0)) * 1024
    lanes = c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64) if c.get("SQ_ACTIVE_INST_VALU") else None
    t = td.get(k, {})
    gui = t.get("GRBM_GUI_ACTIVE", 0.0) / 8.0            # summed over the 8 XCDs
    td_busy = round(t.get("TD_TD_BUSY_sum", 0.0) / 256.0 / gui, 4) if gui else None     # 256 CUs, one TD each
    td_stall = round(t.get("TD_TC_STALL_sum", 0.0) / 256.0 / gui, 4) if gui else None
    rows.append(dict(kernel=k, calls=calls[k], ms_total=round(dur.get(k, 0.0), 4), valu_wave_insts=int(c["SQ_INSTS_VALU"]), salu_insts=int(c["SQ_INSTS_SALU"]),
                     vmem_insts=int(c["SQ_INSTS_VMEM"]), waves=int(c["SQ_WAVES"]), active_lane_frac=None if lanes is None else round(lanes, 4),
                     fetch_kib=int(fe[k].get("FETCH_SIZE", 0)), write_kib=int(wr[k].get("WRITE_SIZE", 0)), hbm_bytes=int(hbm),
                     td_busy_frac=td_busy, td_stalled_on_tc_frac=td_stall))
with open(os.path.join(out, "pmc_per_kernel.csv"), "w") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader(); w.writerows(rows)

dominant = line["roofline"]["kernel"]
dom = [r for r in rows if r["kernel"].startswith(dominant)]
lane_num = sum(sq[r["kernel"]]["SQ_THREAD_CYCLES_VALU"] for r in dom)
lane_den = sum(sq[r["kernel"]]["SQ_ACTIVE_INST_VALU"] for r in dom) * 64


def td_frac(ks):
    """TD busy cycles over GPU-active cycles of the given kernels (time-weighted over their launches)."""
    busy = sum(td.get(k, {}).get("TD_TD_BUSY_sum", 0.0) for k in ks) / 256.0
    gui = sum(td.get(k, {}).get("GRBM_GUI_ACTIVE", 0.0) for k in ks) / 8.0
    return round(busy / gui, 4) if gui else None


sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import device_code_hash      # the hash bench.py compares against: the figures below are only valid for this device code
later = [r["kernel"] for r in dom if r["kernel"].endswith(",false>")]           # wf_trace<COUNT, FIRST>: the rounds after the first
entry = {
    "code_hash": device_code_hash(),
    "dominant_td_busy_frac": td_frac([r["kernel"] for r in dom]),
    "later_rounds_td_busy_frac": td_frac(later) if dominant == "wf_trace" else None,
    "dominant_valu_wave_insts_per_step": sum(r["valu_wave_insts"] for r in dom),
    "dominant_hbm_bytes_per_step": sum(r["hbm_bytes"] for r in dom),
    "dominant_launches_per_step": sum(r["calls"] for r in dom),
    "dominant_ms_per_step_rocprof": round(sum(r["ms_total"] for r in dom), 3),
    "dominant_active_lane_frac": round(lane_num / lane_den, 4) if lane_den else None,
    "all_valu_wave_insts_per_step": sum(r["valu_wave_insts"] for r in rows),
    "all_hbm_bytes_per_step": sum(r["hbm_bytes"] for r in rows),
    "all_ms_per_step_rocprof": round(sum(r["ms_total"] for r in rows), 3),
    "source": f"profiles/{{round}}/roofline_{tag}_pmc_per_kernel.csv (scripts/gpu_roofline_pmc.sh {tag}: rocprofv3 --pmc SQ_* / FETCH_SIZE / WRITE_SIZE passes, --pools 1)",
}
key = line["roofline"]["pmc_key"]
json.dump({key: entry}, open(os.path.join(out, "entry.json"), "w"), indent=1)
print(key)
print(json.dumps(entry, indent=1))
for r in rows:
    print(r)

if "--install" in sys.argv:
    rnd = sys.argv[sys.argv.index("--install") + 1]
    dst = os.path.join("profiles", rnd)
    os.makedirs(dst, exist_ok=True)
    for fn in ("kernel_stats.csv", "pmc_per_kernel.csv", "bench_line.json"):
        shutil.copy(os.path.join(out, fn), os.path.join(dst, f"roofline_{tag}_{fn}"))
    entry["source"] = entry["source"].format(round=rnd)
    pj = os.path.join("profiles", "pmc_counts.json")
    j = json.load(open(pj)) if os.path.exists(pj) else {"_comment": "per-step PMC figures of the dominant kernel and of the whole render, one entry per profiled workload; written by scripts/summarize_roofline_pmc.py --install, read by bench.py (roofline.achieved / traffic)"}
    j[key] = entry
    json.dump(j, open(pj, "w"), indent=1)
    print("installed into", pj)
