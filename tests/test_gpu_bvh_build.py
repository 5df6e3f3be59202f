"""GPU BVH build and rebuild (cgpt_bvh_build / cgpt_bvh_build_ex, all three BuildOptions) against the ORACLE's build (oracle/pt_oracle.c restates BVH.cpp:188-366;
tests/test_oracle_pins.py pins it to the reference's Cube / Duck trees): every 32-byte node word, every tri index, depth and
area equal.  The product's own host build (csrc/host/mesh_bvh.cpp) is compared as well, but it is not the checker."""
import ctypes as C
import time

import numpy as np
import pytest

import oracle as O
import cpugpupathtracing_amd as P
from cpugpupathtracing_amd import _native as N

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def renderer():
    r = P.Renderer(0)
    yield r
    r.close()


def _host_and_gpu(renderer, mesh, option=P.BUILD_SAH_INTERVALS):
    s = P.Scene()
    s.add_material(P.Material())
    s.add_mesh(mesh, 0, option)
    t0 = time.perf_counter()
    host_nodes, host_tri = s.bvh_export(0)
    info = s.bvh_info(0)
    desc = s.flatten()
    obj = desc.objects[0]
    tri_ptr = C.cast(C.addressof(desc.triangles.contents) + obj.tri_offset * C.sizeof(N.Triangle), C.POINTER(N.Triangle))
    t1 = time.perf_counter()
    gpu = renderer.build_bvh(tri_ptr, obj.tri_count, option)
    t2 = time.perf_counter()
    # the checker: the oracle's tree for the same triangles
    o = O.OracleScene()
    o.add_material()
    o.add_mesh(mesh.vertices, mesh.indices, 0, option)              # the three option values are the reference's enum (BVH.h:7-13) everywhere
    on, ot = o.bvh_export(0)
    oi = o.bvh_info(0)
    _assert_same((on, ot, oi.max_depth, oi.total_area), gpu, "oracle")
    return (host_nodes, host_tri, info.max_depth, info.total_area), gpu, t2 - t1


def _assert_same(host, gpu, who="host"):
    hn, ht, hd, ha = host
    gn, gt, gd, ga = gpu
    hn = np.asarray(hn).view(np.uint32).reshape(-1, 8)
    assert gn.shape == hn.shape, (who, gn.shape, hn.shape)
    assert np.array_equal(gt, ht), f"tri order differs from the {who}'s at {np.flatnonzero(gt != ht)[:8]}"
    bad = np.flatnonzero((gn != hn).any(axis=1))
    assert bad.size == 0, f"nodes differ from the {who}'s at {bad[:8]}: gpu {gn[bad[0]]} {who} {hn[bad[0]]}"
    assert gd == hd, (who, gd, hd)
    assert np.float32(ga).tobytes() == np.float32(ha).tobytes(), (who, ga, ha)


@pytest.mark.parametrize("level", [0, 1, 2, 3, 4, 5])
def test_standin_tree_identical(renderer, level):
    host, gpu, _ = _host_and_gpu(renderer, P.Mesh.dragon_standin(level))
    _assert_same(host, gpu)


def test_single_triangle_and_coincident_triangles(renderer):
    v = np.array([[0, 0, 0, 0, 0, 1], [1, 0, 0, 0, 0, 1], [0, 1, 0, 0, 0, 1]], np.float32)
    host, gpu, _ = _host_and_gpu(renderer, P.Mesh.from_arrays(v, np.array([0, 1, 2], np.uint32)))
    _assert_same(host, gpu)
    assert gpu[0].shape[0] == 1
    # 300 copies of one triangle: no plane separates the centroids, so the root stays a leaf
    host, gpu, _ = _host_and_gpu(renderer, P.Mesh.from_arrays(v, np.tile(np.array([0, 1, 2], np.uint32), 300)))
    _assert_same(host, gpu)
    assert gpu[0].shape[0] == 1


@pytest.mark.parametrize("seed,n_tris", [(1, 2), (2, 3), (3, 17), (4, 255), (5, 256), (6, 257), (7, 1000), (8, 5000), (9, 40000)])
def test_random_soup_identical(renderer, seed, n_tris):
    """triangle soup with clustered, duplicated and signed-zero coordinates: ties in the SAH and equal bounds of either zero sign"""
    rng = np.random.default_rng(seed)
    pos = rng.uniform(-4, 4, size=(n_tris * 3, 3)).astype(np.float32)
    pos[rng.random(pos.shape) < 0.15] = 0.0
    pos[rng.random(pos.shape) < 0.10] = -0.0
    pos = np.round(pos * 4) / 4 if seed % 2 else pos             # coarse grid: many equal centroids
    v = np.concatenate([pos, np.tile(np.array([[0, 1, 0]], np.float32), (pos.shape[0], 1))], axis=1).astype(np.float32)
    host, gpu, _ = _host_and_gpu(renderer, P.Mesh.from_arrays(v, np.arange(n_tris * 3, dtype=np.uint32)))
    _assert_same(host, gpu)


def test_large_mesh_identical_and_usable(renderer):
    """level 7 stand-in (327,680 triangles): same tree; uploading the GPU-built arrays in place of the host ones renders the same image"""
    host, gpu, seconds = _host_and_gpu(renderer, P.Mesh.dragon_standin(7))
    _assert_same(host, gpu)
    print(f"GPU build of 327,680 triangles: {seconds * 1e3:.1f} ms, {gpu[0].shape[0]} nodes, depth {gpu[2]}")


def _small_scene(mesh, device_builder):
    s = P.Scene()
    for m in P.REFERENCE_MATERIALS:
        s.add_material(m)
    s.add_mesh(mesh, 3, P.BUILD_SAH_INTERVALS, device_builder=device_builder)
    s.add_light(s.add_sphere((10.0, 10.0, 10.0), 5.0, 2))
    s.set_camera((0, 0, 8), (0, 0, -1), 60.0, 1.0)
    return s


def test_scene_with_device_built_tree_renders_identically(renderer):
    mesh = P.Mesh.dragon_standin(4)
    images = []
    for builder in (None, renderer):
        t0 = time.perf_counter()
        s = _small_scene(mesh, builder)
        dt = time.perf_counter() - t0
        renderer.upload(s)
        renderer.reset_accumulator()
        renderer.reset_stats()
        renderer.render(96, 96, n_samples=4, counters=True)
        images.append((renderer.accumulator().copy(), renderer.stats().tri_tests, s.bvh_export(0), dt))
    assert np.array_equal(images[0][0], images[1][0])
    assert images[0][1] == images[1][1]
    assert np.array_equal(images[0][2][0], images[1][2][0]) and np.array_equal(images[0][2][1], images[1][2][1])


def _soup(seed, n_tris):
    rng = np.random.default_rng(seed)
    pos = rng.uniform(-4, 4, size=(n_tris * 3, 3)).astype(np.float32)
    pos[rng.random(pos.shape) < 0.15] = 0.0
    pos[rng.random(pos.shape) < 0.10] = -0.0
    pos = np.round(pos * 4) / 4 if seed % 2 else pos
    v = np.concatenate([pos, np.tile(np.array([[0, 1, 0]], np.float32), (pos.shape[0], 1))], axis=1).astype(np.float32)
    return P.Mesh.from_arrays(v, np.arange(n_tris * 3, dtype=np.uint32))


@pytest.mark.parametrize("option", [P.BUILD_NAIVE, P.BUILD_SAH_INTERVALS, P.BUILD_SAH_PRIMITIVES])
def test_every_build_option_matches_the_oracle(renderer, option, monkeypatch):
    """BuildOption_NaiveSplit (midpoint of the longest axis, leaves of <= 2), SAHSplitIntervals and SAHSplitPrimitives (never splits,
    SURVEY A-5) on the GPU: the oracle's tree word for word -- stand-in meshes, soups with ties and signed zeros, one triangle,
    and (second pass) with the top levels cut into pieces many levels deep"""
    assert (O.BUILD_NAIVE, O.BUILD_SAH_INTERVALS, O.BUILD_SAH_PRIMITIVES) == (P.BUILD_NAIVE, P.BUILD_SAH_INTERVALS, P.BUILD_SAH_PRIMITIVES)
    for piece_tris in (None, 7):
        if piece_tris is not None:
            monkeypatch.setenv("CGPT_BVH_PIECE_TRIS", str(piece_tris))
        slow = option == P.BUILD_SAH_PRIMITIVES                               # the host / oracle sweep of this option is O(n^2) (ref: BVH.cpp:268-288)
        for level in (0, 2, 3) if slow else (0, 2, 4, 5):
            host, gpu, _ = _host_and_gpu(renderer, P.Mesh.dragon_standin(level), option)
            _assert_same(host, gpu)
            if option == P.BUILD_SAH_PRIMITIVES:
                assert gpu[0].shape[0] == 1 and gpu[2] == 0
            if option == P.BUILD_NAIVE and level >= 2:
                assert gpu[0].shape[0] > 1
        for seed, n_tris in ((2, 1), (3, 2), (4, 3), (5, 17), (6, 257), (7, 1000), (8, 5000), (9, 40000)):
            if slow and n_tris > 1000:
                continue
            host, gpu, _ = _host_and_gpu(renderer, _soup(seed, n_tris), option)
            _assert_same(host, gpu)


@pytest.mark.parametrize("first,then", [(P.BUILD_SAH_INTERVALS, P.BUILD_NAIVE), (P.BUILD_NAIVE, P.BUILD_SAH_INTERVALS), (P.BUILD_SAH_INTERVALS, P.BUILD_SAH_INTERVALS),
                                        (P.BUILD_NAIVE, P.BUILD_SAH_PRIMITIVES)])
def test_device_rebuild_matches_the_oracle_rebuild(renderer, first, then):
    """BVH::Rebuild (ref: BVH.cpp:47-59) does not reset m_tri_indices: the re-split runs over the order the previous build left, and the
    swap partition is order-sensitive.  Device rebuild == oracle rebuild (and != a fresh build where the order matters)."""
    for mesh in (P.Mesh.dragon_standin(4), _soup(7, 3000)):
        s = P.Scene()
        s.add_material(P.Material())
        s.add_mesh(mesh, 0, first, device_builder=renderer)
        o = O.OracleScene()
        o.add_material()
        o.add_mesh(mesh.vertices, mesh.indices, 0, first)
        n0, t0 = s.bvh_export(0)
        on0, ot0 = o.bvh_export(0)
        assert np.array_equal(t0, ot0) and np.array_equal(np.asarray(n0).view(np.uint32), np.asarray(on0).view(np.uint32))
        s.rebuild_bvh(0, then, device_builder=renderer)
        o.rebuild_bvh(0, then)
        n1, t1 = s.bvh_export(0)
        on1, ot1 = o.bvh_export(0)
        assert np.array_equal(t1, ot1), f"triangle order after the rebuild differs at {np.flatnonzero(t1 != ot1)[:8]}"
        assert np.array_equal(np.asarray(n1).view(np.uint32), np.asarray(on1).view(np.uint32))
        i1, oi1 = s.bvh_info(0), o.bvh_info(0)
        assert (i1.nodes_used, i1.max_depth) == (oi1.nodes_used, oi1.max_depth)
        assert np.float32(i1.total_area).tobytes() == np.float32(oi1.total_area).tobytes()      # Rebuild leaves m_total_area alone
        # and the rebuilt tree renders
        if then != P.BUILD_SAH_PRIMITIVES:
            s.add_light(s.add_sphere((10.0, 10.0, 10.0), 5.0, 0))
            s.set_camera((0, 0, 8), (0, 0, -1), 60.0, 1.0)
            renderer.upload(s)
            renderer.render(32, 32, 1)


def test_device_build_input_errors(renderer):
    s = P.Scene()
    s.add_material(P.Material())
    s.add_mesh(P.Mesh.dragon_standin(1), 0, P.BUILD_SAH_INTERVALS)
    desc = s.flatten()
    with pytest.raises(P.DeviceError, match="unknown build option"):
        renderer.build_bvh(desc.triangles, desc.objects[0].tri_count, 7)
    bad = np.zeros(desc.objects[0].tri_count, np.uint32)                  # not a permutation
    with pytest.raises(P.DeviceError, match="not a permutation"):
        renderer.build_bvh(desc.triangles, desc.objects[0].tri_count, P.BUILD_NAIVE, bad)


def test_build_time_host_vs_device(renderer):
    mesh = P.Mesh.dragon_standin(7)
    t = []
    for builder in (None, renderer, renderer):
        t0 = time.perf_counter(); _small_scene(mesh, builder); t.append(time.perf_counter() - t0)
    print(f"327,680 triangles: host build {t[0] * 1e3:.0f} ms, device build {t[2] * 1e3:.0f} ms (first call {t[1] * 1e3:.0f} ms)")


@pytest.mark.parametrize("piece_tris,wave_tris,quarter_tris", [(1, 2048, 32), (7, 0, 0), (64, 2048, 0), (10 ** 9, 0, 0), (10 ** 9, 10 ** 9, 0), (10 ** 9, 10 ** 9, 10 ** 9)])
def test_every_level_strategy_on_small_meshes(renderer, monkeypatch, piece_tris, wave_tris, quarter_tris):
    """The top levels of the tree are built by many workgroups per node (bvh_build.hip: wide_*), their partial results combined in
    piece order; the levels below by one workgroup per node; the deep levels by one wavefront per node.  CGPT_BVH_PIECE_TRIS shrinks
    the pieces so that small meshes take the first path many levels deep, down to pieces of a single triangle; CGPT_BVH_WAVE_TRIS
    moves the border between the other two (0: workgroups only, huge: wavefronts only), CGPT_BVH_QUARTER_TRIS the one to a quarter
    wavefront per node (huge: every node).  The same trees as the oracle's in every
    combination, signed zeros and SAH ties included."""
    monkeypatch.setenv("CGPT_BVH_PIECE_TRIS", str(piece_tris))
    monkeypatch.setenv("CGPT_BVH_WAVE_TRIS", str(wave_tris))
    monkeypatch.setenv("CGPT_BVH_QUARTER_TRIS", str(quarter_tris))
    for level in (1, 3, 4):
        host, gpu, _ = _host_and_gpu(renderer, P.Mesh.dragon_standin(level))
        _assert_same(host, gpu)
    for seed, n_tris in ((3, 17), (5, 256), (6, 257), (7, 1000), (8, 5000)):
        rng = np.random.default_rng(seed)
        pos = rng.uniform(-4, 4, size=(n_tris * 3, 3)).astype(np.float32)
        pos[rng.random(pos.shape) < 0.15] = 0.0
        pos[rng.random(pos.shape) < 0.10] = -0.0
        pos = np.round(pos * 4) / 4 if seed % 2 else pos
        v = np.concatenate([pos, np.tile(np.array([[0, 1, 0]], np.float32), (pos.shape[0], 1))], axis=1).astype(np.float32)
        host, gpu, _ = _host_and_gpu(renderer, P.Mesh.from_arrays(v, np.arange(n_tris * 3, dtype=np.uint32)))
        _assert_same(host, gpu)
