#!/usr/bin/env python3
"""Sums rocprofv3 counter_collection.csv values per kernel (non-COUNT variants of the cgpt kernels)."""
import collections, csv, glob, os, sys
dirs = sys.argv[1:] or ["gpurun_out/pmc_a", "gpurun_out/pmc_b", "gpurun_out/pmc_c"]
agg = collections.defaultdict(float)
for d in dirs:
    fs = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))
    if not fs: continue
    for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("cgpt::", "")
        if "wf_" not in k and "megakernel" not in k: continue
        if "<true>" in k: continue
        agg[(k, r["Counter_Name"])] += float(r["Counter_Value"])
kernels = sorted({k for k, _ in agg})
for k in kernels:
    c = {n: v for (kk, n), v in agg.items() if kk == k}
    print(k)
    print("   ", " ".join(f"{n}={v:.4g}" for n, v in sorted(c.items())))
    if "SQ_THREAD_CYCLES_VALU" in c and c.get("SQ_ACTIVE_INST_VALU"):
        print(f"    active lanes per VALU inst: {c['SQ_THREAD_CYCLES_VALU'] / (c['SQ_ACTIVE_INST_VALU'] * 64):.2f}")
    if c.get("SQ_WAVE_CYCLES") and "SQ_WAIT_ANY" in c:
        print(f"    wave time: wait_any {c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']:.2f} active {c['SQ_ACTIVE_INST_ANY'] / c['SQ_WAVE_CYCLES']:.2f} wait_inst {c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']:.2f}")
    if c.get("TCC_HIT_sum"):
        print(f"    L2 hit rate {c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']):.2f}  L1 hit rate {1 - c['TCP_TCC_READ_REQ_sum'] / c['TCP_TOTAL_CACHE_ACCESSES_sum']:.2f}")
