#!/usr/bin/env python3
"""On the GPU box, with the -DCGPT_STEP_MAP diagnostic build (CGPT_LIB_PATH=.../libcpugpupt_stepmap.so): dependent record fetches
(inner steps + triangle tests) per path of a one-sample frame -- the length distribution of the chains that bound a one-sample call."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cpugpupathtracing_amd as P
W, H = 1920, 1080
r = P.Renderer(0)
r.upload(P.Scene.reference_layout(P.Mesh.dragon_standin(6), 3, W / H, P.BUILD_SAH_INTERVALS))
for seed in (0x12345678, 7):
    r.reset_accumulator()
    r.render(W, H, 1, seed=seed, kernel=P.KERNEL_MEGAKERNEL, counters=True)
    steps = r.pixels().astype(np.int64)
    st = r.stats()
    flat = np.sort(steps.ravel())[::-1]
    print(f"seed {seed:#x}: paths {steps.size}, fetches total {flat.sum()}, mean {flat.mean():.1f}, max {flat[0]}, top10 {flat[:10].tolist()}")
    print("  percentiles 50/90/99/99.9/99.99:", [int(np.percentile(flat, q)) for q in (50, 90, 99, 99.9, 99.99)])
    tiles = steps[:H - H % 8, :W - W % 8].reshape(H // 8, 8, W // 8, 8)
    tmax = tiles.max(axis=(1, 3)); tsum = tiles.sum(axis=(1, 3))
    print(f"  8x8 tiles: max-of-tile mean {tmax.mean():.0f}, sum-of-tile mean {tsum.mean():.0f}; lane efficiency of a tile-per-wave kernel = sum/(64*max) = {tsum.sum() / (64.0 * tmax.sum()):.3f}")
    rows = steps.max(axis=1)
    print(f"  longest chain per image row: max {rows.max()} at row {rows.argmax()}, row 540: {rows[540]}")
    r.reset_stats()
