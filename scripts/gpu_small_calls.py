#!/usr/bin/env python3
"""On the GPU box: ms per cgpt_render call for small calls over several frame sizes and kernels (the AUTO thresholds of
cgpt_abi.hip: RenderEnqueue are read off this table).  usage: python scripts/gpu_small_calls.py [samples=1,2,4,8] [sizes=64x64,256x256,960x540,1920x1080] [knob=value ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cpugpupathtracing_amd as P

knobs = {a.split("=")[0]: a.split("=")[1] for a in sys.argv[1:] if "=" in a}
samples = [int(x) for x in knobs.pop("samples", "1,2,4,8").split(",")]
sizes = [tuple(int(v) for v in s.split("x")) for s in knobs.pop("sizes", "64x64,256x256,960x540,1920x1080").split(",")]
knobs = {k: int(v) for k, v in knobs.items()}
mesh = P.Mesh.dragon_standin(6)
kernels = ((P.KERNEL_MEGAKERNEL, "megakernel"), (P.KERNEL_PERSISTENT, "persistent"), (P.KERNEL_WAVEFRONT, "wavefront"), (P.KERNEL_AUTO, "auto"))
print(f"# ms per cgpt_render call (device time, mean of 20 calls after 2 warm-up calls); glass stand-in level 6; knobs {knobs}")
print("# size      samples " + " ".join(f"{n:>11s}" for _, n in kernels) + "   auto ran")
for W, H in sizes:
    r = P.Renderer(0)
    r.upload(P.Scene.reference_layout(mesh, 3, W / H, P.BUILD_SAH_INTERVALS))
    if knobs:
        r.set_tuning(**knobs)
    for n in samples:
        row, ran = [], 0
        for k, name in kernels:
            r.reset_accumulator()
            r.render(W, H, n, kernel=k); r.render(W, H, n, kernel=k)
            r.reset_stats()
            calls = 20
            for _ in range(calls):
                r.render(W, H, n, kernel=k)
            st = r.stats()
            row.append(st.kernel_ms / calls)
            ran = st.last_kernel
        print(f"{W:5d}x{H:<5d} {n:6d} " + " ".join(f"{v:11.3f}" for v in row) + f"   {ran}", flush=True)
    r.close()
