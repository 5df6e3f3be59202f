#!/usr/bin/env python3
"""On the GPU box: step statistics of the persistent kernel's COUNT instantiation (wave steps and lanes per step by state) for one render.
usage: CGPT_WF_PROFILE=1 python scripts/gpu_pt_profile.py [W H spp] [knob=value ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cpugpupathtracing_amd as P
args = [a for a in sys.argv[1:] if "=" not in a]
knobs = {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[1:] if "=" in a}
W, H, spp = (int(args[0]), int(args[1]), int(args[2])) if len(args) >= 3 else (1920, 1080, 64)
r = P.Renderer(0)
r.upload(P.Scene.reference_layout(P.Mesh.dragon_standin(6), 3, W / H, P.BUILD_SAH_INTERVALS))
if knobs:
    r.set_tuning(**knobs)
r.render(W, H, spp, kernel=P.KERNEL_PERSISTENT)
r.reset_stats()
r.render(W, H, spp, kernel=P.KERNEL_PERSISTENT, counters=True)
st = r.stats()
print(f"{W}x{H}x{spp} knobs {knobs}: {st.kernel_ms:.2f} ms (COUNT kernel), rays {st.traced_rays}, inner {st.inner_steps}, tris {st.tri_tests}", flush=True)
r.reset_stats()
r.render(W, H, spp, kernel=P.KERNEL_PERSISTENT)
print(f"  production kernel: {r.stats().kernel_ms:.2f} ms", flush=True)
