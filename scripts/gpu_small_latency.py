#!/usr/bin/env python3
"""On the GPU box: device time of tiny renders (a few pixels in the middle of the glass stand-in), per kernel: the latency floor
of one path / one 64-path tile, which bounds a one-sample frame from below."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cpugpupathtracing_amd as P
W, H = 1920, 1080
r = P.Renderer(0)
r.upload(P.Scene.reference_layout(P.Mesh.dragon_standin(6), 3, W / H, P.BUILD_SAH_INTERVALS))
for name, kernel in (("megakernel", P.KERNEL_MEGAKERNEL), ("wavefront", P.KERNEL_WAVEFRONT), ("persistent", P.KERNEL_PERSISTENT)):
    for rows, spp in (((540, 541), 1), ((536, 544), 1), ((536, 544), 8), ((512, 576), 1), ((0, 8), 1)):
        r.render(W, H, spp, rows=rows, kernel=kernel)
        r.reset_stats()
        n = 10
        for _ in range(n):
            r.render(W, H, spp, rows=rows, kernel=kernel)
        st = r.stats()
        print(f"{name:10s} rows {rows} x {W} px, {spp} spp: {st.kernel_ms / n * 1e3:8.1f} us/call, {st.traced_rays / n:9.0f} rays/call", flush=True)
