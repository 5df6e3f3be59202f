#!/usr/bin/env python3
"""Here, after `gpurun -- bash scripts/gpu_final_profiles.sh <round>` + `bench.py` + `scripts/gpu_final_tables.sh`: copies the summaries gpurun merged into
gpurun_out/ to profiles/<round>/ (and profiles/pmc_counts.json), and rebuilds the configuration table.  usage: python scripts/install_final.py r03"""
import glob, os, shutil, subprocess, sys
rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
out = os.path.join("profiles", rnd)
os.makedirs(out, exist_ok=True)
for f in glob.glob(f"gpurun_out/profiles_{rnd}/roofline_*"):
    shutil.copy(f, out)
shutil.copy(f"gpurun_out/profiles_{rnd}/pmc_counts.json", "profiles/pmc_counts.json")
pairs = {"final_bench_default.json": "bench_default.json", "final_bench_persistent.json": "bench_persistent.json", "final_bench_comparison.json": "bench_comparison.json",
         "final_bench_comparison_pt.json": "bench_comparison_pt.json", "final_bench_brute.json": "bench_brute.json", "default_trace_kernel_stats.csv": "bench_default_kernel_stats.csv",
         "final_frame_time.txt": "frame_time_1080p.txt", "final_small_calls.txt": "small_calls_table.txt", "final_small_latency.txt": "small_call_latency.txt",
         "final_wave_time.txt": "wave_time_breakdown.txt", "final_mem_path.txt": "mem_path_pmc.txt", "timeline_libcpugpupt.txt": "production_timeline.txt"}
for src, dst in pairs.items():
    p = os.path.join("gpurun_out", src)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join(out, dst))
    else:
        print("missing:", p)
if os.path.exists("gpurun_out/configs.jsonl"):
    subprocess.check_call([sys.executable, "scripts/configs_table.py", rnd])
