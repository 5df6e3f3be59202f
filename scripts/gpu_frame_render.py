#!/usr/bin/env python3
"""Renders a few frames of n samples per call with one kernel (for rocprofv3 timelines of the small-call case)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cpugpupathtracing_amd as P
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
kernel = {"wavefront": P.KERNEL_WAVEFRONT, "megakernel": P.KERNEL_MEGAKERNEL, "auto": P.KERNEL_AUTO, "persistent": P.KERNEL_PERSISTENT}[sys.argv[2] if len(sys.argv) > 2 else "wavefront"]
W, H = 1920, 1080
r = P.Renderer(0)
r.upload(P.Scene.reference_layout(P.Mesh.dragon_standin(6), 3, W / H, P.BUILD_SAH_INTERVALS))
for _ in range(6):
    r.render(W, H, n, kernel=kernel)
print(r.stats().kernel_ms / 6)
