"""Checks the CPU oracle against outputs of the verbatim reference recorded in SURVEY.md section 8c.

These are NOT pins in the rules' sense (parity stays "unpinned": the numbers come from the survey's one-off build with stand-in
headers, which cannot be re-run here, and the reference itself holds no fixtures) -- they are the strongest anchor available.
The reference cannot be built in this image (DESIGN.md "Oracle"), so these recorded vectors -- BVH statistics of the
reference's Cube and Duck assets for all three build options, areas, and the 4- / 16-frame Duck render in the
reference's own RNG/tile order -- are what anchor the oracle.  They need the reference's asset files and therefore run
only where /root/reference exists (not on the GPU box).
"""
import os

import numpy as np
import pytest

import oracle as O

# (tris, area) and per build option (nodes used, leaves, max leaf, depth): SURVEY 8c "Pins captured by the survey"
PINS = {
    "Cube/Cube.gltf": dict(tris=12, area=24.0, bvh={0: (3, 2, 6, 1), 1: (19, 10, 2, 6), 2: (1, 1, 12, 0)}),
    "Duck/Duck.gltf": dict(tris=4212, area=70235.156250, verts=2399, indices=12636,
                           bvh={0: (4927, 2464, 9, 21), 1: (8421, 4211, 2, 16), 2: (1, 1, 4212, 0)}),
}


@pytest.mark.parametrize("asset", sorted(PINS))
def test_bvh_statistics_match_reference(reference_assets, asset):
    pin = PINS[asset]
    v, i = O.load_gltf_reference_semantics(os.path.join(reference_assets, asset))
    if "verts" in pin:
        assert (v.shape[0], i.size) == (pin["verts"], pin["indices"])
    for opt, want in pin["bvh"].items():
        s = O.OracleScene()
        s.add_material()
        s.add_mesh(v, i, 0, opt)
        b = s.bvh_info(0)
        assert b.num_triangles == pin["tris"]
        assert (b.nodes_used, b.num_leaves, b.max_leaf_size, b.max_depth) == want
        assert b.total_area == np.float32(pin["area"])


def test_ground_quad_single_leaf():
    # ref: Main.cpp:789-800 -> 1 node (leaf of 2), area 4 000 000
    s = O.OracleScene()
    s.add_material()
    gv = np.array([[-1000, -3, 1000, 0, 1, 0], [-1000, -3, -1000, 0, 1, 0], [1000, -3, -1000, 0, 1, 0], [1000, -3, 1000, 0, 1, 0]], np.float32)
    s.add_mesh(gv, [0, 1, 2, 2, 3, 0], 0, O.BUILD_SAH_INTERVALS)
    b = s.bvh_info(0)
    assert (b.nodes_used, b.max_leaf_size, b.total_area) == (1, 2, 4000000.0)


@pytest.mark.parametrize("frames,rays,acc_sum,centre", [(4, 274373, 1555.643398, 0xFF190708), (16, 1097314, 6219.162458, 0xFF150606)])
def test_duck_render_matches_reference(reference_assets, frames, rays, acc_sum, centre):
    """Duck, 256x256, camera (0,80,400)->-z fov 60, blue diffuse, one sphere light, ADVANCED, defaults, serial tile order,
    three xorshift streams @0x12345678 (SURVEY 8c last pin row)."""
    v, i = O.load_gltf_reference_semantics(os.path.join(reference_assets, "Duck/Duck.gltf"))
    s = O.OracleScene()
    s.add_material(albedo=(0.2, 0.2, 0.8))
    s.add_material(albedo=(1, 1, 1))
    s.add_material(emissive=(1.0, 0.95, 0.8), intensity=10.0, is_light=True)
    s.add_mesh(v, i, 0, O.BUILD_SAH_INTERVALS)
    s.add_light(s.add_sphere((300, 400, 300), 100.0, 2))
    s.set_camera((0, 80, 400), (0, 0, -1), 60.0, 1.0)
    s.render(256, 256, frames, O.MODE_ADVANCED, O.DEBUG_NONE, O.RNG_REFERENCE_XORSHIFT)
    assert s.stats().traced_rays == rays
    acc = s.accumulator()
    per_pixel = (acc[..., 0] + acc[..., 1]) + acc[..., 2]          # float sum per pixel, then a double total
    assert f"{float(per_pixel.astype(np.float64).sum()):.6f}" == f"{acc_sum:.6f}"
    assert int(s.pixels()[128, 128]) == centre
    assert s.num_accumulated() == frames


def test_reference_rng_mode_needs_multiples_of_16():
    # SURVEY A-1: the verbatim tile loop is only valid for W%16==0 and H%16==0
    s = O.OracleScene()
    s.add_material(albedo=(1, 1, 1))
    s.add_sphere((0, 0, -5), 1.0, 0)
    with pytest.raises(ValueError):
        s.render(40, 40, 1, O.MODE_ADVANCED, O.DEBUG_NONE, O.RNG_REFERENCE_XORSHIFT)
