"""Committed golden vectors (tests/golden, made by the pinned oracle): the oracle must still reproduce them (CPU),
and the HIP path must match them through the C ABI (GPU)."""
import os

import numpy as np
import pytest

import oracle as O
import cpugpupathtracing_amd as P
from scenes import MAT_SPEC_DIFFUSE, reference_layout_pair, rmse

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "standin_l2_golden.npz"))
W, H, SPP, SEED = int(G["W"]), int(G["H"]), int(G["spp"]), int(G["seed"])
CASES = [("diffuse", 1, True), ("specdiffuse", 4, True), ("glass", 3, False)]


@pytest.mark.parametrize("name,mat,exact", CASES)
def test_oracle_reproduces_golden(name, mat, exact):
    o, _ = reference_layout_pair(G["vertices"], G["indices"], mat, extra_materials=(MAT_SPEC_DIFFUSE,))
    o.render(W, H, SPP, O.MODE_ADVANCED, O.DEBUG_NONE, O.RNG_PIXEL_PCG, SEED, nthreads=3)
    assert np.array_equal(o.accumulator().view(np.uint32), G[f"acc_{name}"].view(np.uint32))
    st = o.stats()
    assert [st.traced_rays, st.inner_steps, st.tri_tests, st.bvh_depth_sum, st.closest_hits] == list(G[f"counters_{name}"])


@pytest.mark.gpu
@pytest.mark.parametrize("name,mat,exact", CASES)
def test_hip_matches_golden(name, mat, exact):
    _, s = reference_layout_pair(G["vertices"], G["indices"], mat, extra_materials=(MAT_SPEC_DIFFUSE,))
    r = P.Renderer(0)
    r.upload(s)
    r.render(W, H, SPP, seed=SEED, counters=True)
    acc = r.accumulator()
    assert rmse(acc[..., :3] / SPP, G[f"acc_{name}"][..., :3] / SPP) < 1e-4
    st = r.stats()
    assert [st.traced_rays, st.inner_steps, st.tri_tests, st.bvh_depth_sum, st.closest_hits] == list(G[f"counters_{name}"])
    if exact:
        assert np.array_equal(acc.view(np.uint32), G[f"acc_{name}"].view(np.uint32))
        assert np.array_equal(r.pixels(), G[f"pixels_{name}"])
    r.close()


@pytest.mark.gpu
def test_hip_hit_records_match_golden():
    _, s = reference_layout_pair(G["vertices"], G["indices"], 1, extra_materials=(MAT_SPEC_DIFFUSE,))
    r = P.Renderer(0)
    r.upload(s)
    t, obj, tri, dep = r.intersect_rays(G["ray_o"], G["ray_d"])
    assert np.array_equal(obj, G["hit_obj"]) and np.array_equal(t.view(np.uint32), G["hit_t"].view(np.uint32))
    hit = obj != 0xFFFFFFFF
    assert np.array_equal(tri[hit], G["hit_tri"][hit]) and np.array_equal(dep, G["hit_depth"])
    r.close()
