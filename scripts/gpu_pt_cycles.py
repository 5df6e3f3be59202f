#!/usr/bin/env python3
"""On the GPU box with the -DCGPT_PHASE_CYCLES build: where the persistent kernel's waves spend their cycles for n-sample 1080p calls.
usage: CGPT_WF_PROFILE=1 CGPT_LIB_PATH=.../libcpugpupt_cyc.so python scripts/gpu_pt_cycles.py [samples=1,4,64] [knob=value ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cpugpupathtracing_amd as P
knobs = {a.split("=")[0]: a.split("=")[1] for a in sys.argv[1:] if "=" in a}
samples = [int(x) for x in knobs.pop("samples", "1,4,64").split(",")]
mat = int(knobs.pop("mat", 3))
knobs = {k: int(v) for k, v in knobs.items()}
W, H = 1920, 1080
r = P.Renderer(0)
r.upload(P.Scene.reference_layout(P.Mesh.dragon_standin(6), mat, W / H, P.BUILD_SAH_INTERVALS))
if knobs:
    r.set_tuning(**knobs)
for n in samples:
    r.render(W, H, n, kernel=P.KERNEL_PERSISTENT)
    r.reset_stats()
    print(f"--- n_samples={n}", file=sys.stderr, flush=True)
    r.render(W, H, n, kernel=P.KERNEL_PERSISTENT)
    print(f"n_samples={n}: {r.stats().kernel_ms:.3f} ms", file=sys.stderr, flush=True)
