#!/usr/bin/env python3
"""Generates tests/golden/standin_l2_golden.npz with the CPU oracle (pinned to the reference, tests/test_oracle_pins.py).

The reference itself cannot be run here (DESIGN.md "Oracle"), so the golden vectors are produced by the pinned oracle
on a synthetic scene: inputs (mesh arrays) and expected outputs (hit records of the primary rays, 48x48 float4
accumulators after 4 samples for three materials, ray/step counters).  Re-run: python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))

import oracle as O                                   # noqa: E402
from scenes import MAT_SPEC_DIFFUSE, reference_layout_pair, standin_mesh   # noqa: E402

W = H = 48
SPP = 4
SEED = 0x12345678


def main():
    v, i = standin_mesh(2)
    out = {"vertices": v, "indices": i, "W": W, "H": H, "spp": SPP, "seed": SEED}
    for name, mat in (("diffuse", 1), ("specdiffuse", 4), ("glass", 3)):
        o, _ = reference_layout_pair(v, i, mat, extra_materials=(MAT_SPEC_DIFFUSE,))
        o.render(W, H, SPP, O.MODE_ADVANCED, O.DEBUG_NONE, O.RNG_PIXEL_PCG, SEED, nthreads=1)
        st = o.stats()
        out[f"acc_{name}"] = o.accumulator()
        out[f"pixels_{name}"] = o.pixels()
        out[f"counters_{name}"] = np.array([st.traced_rays, st.inner_steps, st.tri_tests, st.bvh_depth_sum, st.closest_hits], np.uint64)
        if name == "diffuse":
            ro, rd = o.camera_rays(W, H)
            t, obj, tri, dep = o.intersect_rays(ro.reshape(-1, 3), rd.reshape(-1, 3))
            out.update(ray_o=ro.reshape(-1, 3), ray_d=rd.reshape(-1, 3), hit_t=t, hit_obj=obj, hit_tri=tri, hit_depth=dep)
    np.savez_compressed(os.path.join(HERE, "standin_l2_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "standin_l2_golden.npz"))


if __name__ == "__main__":
    main()
