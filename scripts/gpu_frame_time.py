#!/usr/bin/env python3
"""On the GPU box: ms per cgpt_render call for small sample counts per call (the reference's main loop renders ONE sample per
Render(), ref: Main.cpp:702,825-942), both kernels.  usage: python scripts/gpu_frame_time.py [W H] [samples=1,2,4,8] [level=6] [knob=value ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cpugpupathtracing_amd as P

args = [a for a in sys.argv[1:] if "=" not in a]
knobs = {a.split("=")[0]: a.split("=")[1] for a in sys.argv[1:] if "=" in a}
samples = [int(x) for x in knobs.pop("samples", "1,2,4,8").split(",")]
level = int(knobs.pop("level", 6))
knobs = {k: int(v) for k, v in knobs.items()}
W, H = (int(args[0]), int(args[1])) if len(args) >= 2 else (1920, 1080)
mesh = P.Mesh.dragon_standin(level)
r = P.Renderer(0)
r.upload(P.Scene.reference_layout(mesh, 3, W / H, P.BUILD_SAH_INTERVALS))
if knobs:
    r.set_tuning(**knobs)
for kernel, name in ((P.KERNEL_MEGAKERNEL, "megakernel"), (P.KERNEL_WAVEFRONT, "wavefront"), (P.KERNEL_PERSISTENT, "persistent"), (P.KERNEL_AUTO, "auto")):
    for n in samples:
        r.reset_accumulator()
        try:
            r.render(W, H, n, kernel=kernel)       # warm: allocations
        except P.DeviceError as e:                 # an older build of the library without this kernel (A/B runs with CGPT_LIB_PATH)
            print(f"{W}x{H} {name:10s} n_samples={n}: not in this build ({e})", flush=True)
            continue
        r.render(W, H, n, kernel=kernel)
        r.reset_stats()
        t0 = time.perf_counter()
        calls = 20 if n <= 8 else 5
        for _ in range(calls):
            r.render(W, H, n, kernel=kernel)
        dt = (time.perf_counter() - t0) / calls * 1e3
        st = r.stats()
        print(f"{W}x{H} {name:10s} n_samples={n}: {dt:7.3f} ms/call host, {st.kernel_ms / calls:7.3f} ms/call device, "
              f"{dt / n:6.3f} ms/frame, {st.traced_rays / calls / dt / 1e3:8.1f} Mrays/s, launches/call {st.kernel_launches / calls:.0f}", flush=True)
