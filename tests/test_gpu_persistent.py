"""The persistent path kernel (CGPT_KERNEL_PERSISTENT: one launch, lanes own paths) against the oracle and the other two
render paths.  tests/test_gpu_parity.py runs the shared scene / edge-case tests over all three kernels; this file holds what is
specific to this one: batches and double buffering, continuation, bands, debug views, settings, the reference's own call
pattern (one sample per call)."""
import numpy as np
import pytest

import oracle as O
import cpugpupathtracing_amd as P
from scenes import MAT_SPEC_DIFFUSE, reference_layout_pair, rmse, standin_mesh

pytestmark = pytest.mark.gpu
RMSE_TOL = 1e-4
K = P.KERNEL_PERSISTENT


@pytest.fixture(scope="module")
def renderer():
    r = P.Renderer(0)
    yield r
    r.close()


@pytest.mark.parametrize("mat,exact", [(1, True), (4, True), (3, False)])
def test_matches_oracle_and_the_other_kernels(renderer, mat, exact):
    v, i = standin_mesh(3)
    o, s = reference_layout_pair(v, i, mat, extra_materials=(MAT_SPEC_DIFFUSE,))
    W, H, spp = 100, 72, 19
    o.render(W, H, spp, O.MODE_ADVANCED, O.DEBUG_NONE, O.RNG_PIXEL_PCG, 0x12345678, nthreads=8)
    renderer.upload(s)
    renderer.reset_accumulator(); renderer.reset_stats()
    renderer.render(W, H, spp, kernel=K, counters=True)
    a0, a1 = o.accumulator(), renderer.accumulator()
    so, sg = o.stats(), renderer.stats()
    assert np.array_equal(a0[..., 3], a1[..., 3])
    assert rmse(a0[..., :3] / spp, a1[..., :3] / spp) < RMSE_TOL
    assert (so.traced_rays, so.inner_steps, so.tri_tests, so.bvh_depth_sum, so.closest_hits) == \
           (sg.traced_rays, sg.inner_steps, sg.tri_tests, sg.bvh_depth_sum, sg.closest_hits)
    assert abs(so.total_energy_received - sg.total_energy_received) < 1e-6 * max(1.0, so.total_energy_received)
    if exact:
        assert np.array_equal(a0.view(np.uint32), a1.view(np.uint32))
        assert np.array_equal(o.pixels(), renderer.pixels())
    px = renderer.pixels()
    for other in (P.KERNEL_MEGAKERNEL, P.KERNEL_WAVEFRONT):
        renderer.reset_accumulator()
        renderer.render(W, H, spp, kernel=other)
        assert np.array_equal(renderer.accumulator().view(np.uint32), a1.view(np.uint32))      # same bits on all three GPU paths, glass included
        assert np.array_equal(renderer.pixels(), px)


def test_batches_double_buffering_and_continuation():
    """a 1 Mi-path batch limit forces many batches on two alternating buffers; samples are still added in order"""
    v, i = standin_mesh(3)
    o, s = reference_layout_pair(v, i, 4, aspect=2.0, extra_materials=(MAT_SPEC_DIFFUSE,))
    W, H, spp = 512, 256, 37                      # 131 072 pixels: 8 samples per batch -> 5 batches
    r = P.Renderer(0)
    r.upload(s)
    r.render(W, H, spp, seed=7, kernel=K)
    ref = r.accumulator().copy()
    rays = r.stats().traced_rays
    for knobs in ({"pt_max_paths_mi": 1}, {"pt_max_paths_mi": 1, "pt_streams": 1}, {"pt_max_paths_mi": 3, "pt_refill": 1, "pt_inner_repeat": 1, "pt_shade_shift": 2},
                  {"pt_path_order": 0}, {"pt_path_order": 1}, {"pt_max_paths_mi": 1, "pt_path_order": 0, "pt_chunk": 3, "pt_fine_rounds": 0},
                  {"pt_max_paths_mi": 2, "pt_path_order": 1, "pt_chunk": 5}):
        q = P.Renderer(0)
        q.upload(s)
        q.set_tuning(**knobs)
        q.render(W, H, 20, seed=7, kernel=K)
        q.render(W, H, 17, seed=7, kernel=K)      # continues at sample 20 (data.num_accumulated, ref: Main.cpp:702)
        assert q.stats().traced_rays == rays
        assert np.array_equal(q.accumulator().view(np.uint32), ref.view(np.uint32)), knobs
        q.close()
    o.render(W, H, 2, O.MODE_ADVANCED, O.DEBUG_NONE, O.RNG_PIXEL_PCG, 7, nthreads=8)
    r.reset_accumulator()
    r.render(W, H, 2, seed=7, kernel=K)
    assert np.array_equal(r.accumulator().view(np.uint32), o.accumulator().view(np.uint32))
    r.close()


def test_settings_debug_views_and_bands(renderer):
    v, i = standin_mesh(2)
    for settings in (P.Settings(next_event_estimation_enabled=False), P.Settings(max_ray_depth=0),
                     P.Settings(russian_roulette_enabled=False, cosine_weighted_diffuse_reflection_enabled=False, max_ray_depth=2)):
        o, s = reference_layout_pair(v, i, 4, extra_materials=(MAT_SPEC_DIFFUSE,), settings=settings)
        o.render(64, 40, 3, O.MODE_ADVANCED, O.DEBUG_NONE, O.RNG_PIXEL_PCG, 0x12345678, nthreads=4)
        renderer.upload(s); renderer.reset_accumulator(); renderer.reset_stats()
        renderer.render(64, 40, 3, kernel=K)
        assert np.array_equal(o.accumulator().view(np.uint32), renderer.accumulator().view(np.uint32))
        assert o.stats().traced_rays == renderer.stats().traced_rays
    for debug in (P.DEBUG_RAY_DEPTH, P.DEBUG_BVH_DEPTH):
        for mode in (P.MODE_ADVANCED, P.MODE_BRUTE_FORCE, P.MODE_COMPARISON):
            st = P.Settings(debug_render_mode=debug, render_mode=mode)
            o, s = reference_layout_pair(v, i, 3, settings=st)
            o.render(64, 64, 1, mode, debug, O.RNG_PIXEL_PCG, 3, nthreads=2)
            renderer.upload(s); renderer.reset_accumulator()
            renderer.render(64, 64, 1, seed=3, kernel=K)
            assert np.array_equal(renderer.pixels(), o.pixels()), (debug, mode)
    o, s = reference_layout_pair(v, i, 3, aspect=96 / 64)
    renderer.upload(s); renderer.reset_accumulator()
    renderer.render(96, 64, 3, seed=5, kernel=K)
    full = renderer.accumulator()
    renderer.render(96, 64, 3, seed=5, rows=(20, 41), kernel=K)
    assert np.array_equal(renderer.accumulator().view(np.uint32), full[20:41].view(np.uint32))
    renderer.render(96, 64, 3, seed=5, interleave=(4, 3, 2), kernel=K)
    from cpugpupathtracing_amd import distributed as D
    assert np.array_equal(renderer.accumulator().view(np.uint32), full[D.interleaved_rows(64, 2, 3, 4)].view(np.uint32))


@pytest.mark.parametrize("W,H,spp", [(1, 1, 1), (7, 3, 5), (65, 9, 2), (130, 70, 3)])
def test_odd_sizes_equal_the_megakernel(renderer, W, H, spp):
    v, i = standin_mesh(2)
    _, s = reference_layout_pair(v, i, 3, aspect=W / H)
    renderer.upload(s)
    renderer.reset_accumulator()
    renderer.render(W, H, spp, seed=11, kernel=P.KERNEL_MEGAKERNEL)
    want = renderer.accumulator().copy()
    renderer.reset_accumulator()
    renderer.render(W, H, spp, seed=11, kernel=K)
    assert np.array_equal(renderer.accumulator().view(np.uint32), want.view(np.uint32))


def test_one_sample_per_call_like_the_reference_main_loop(renderer):
    """the reference renders ONE sample per Render() (ref: Main.cpp:702,825-942): 12 calls of 1 sample == one call of 12, and AUTO
    (which picks the kernel for the call size) gives the same bits"""
    v, i = standin_mesh(3)
    o, s = reference_layout_pair(v, i, 3, aspect=160 / 90)
    o.render(160, 90, 12, O.MODE_ADVANCED, O.DEBUG_NONE, O.RNG_PIXEL_PCG, 21, nthreads=8)
    renderer.upload(s)
    for kernel in (K, P.KERNEL_AUTO):
        renderer.reset_accumulator()
        for _ in range(12):
            renderer.render(160, 90, 1, seed=21, kernel=kernel)
        got = renderer.accumulator().copy()
        assert renderer.num_accumulated == 12
        assert rmse(got[..., :3] / 12, o.accumulator()[..., :3] / 12) < RMSE_TOL
        renderer.reset_accumulator()
        renderer.render(160, 90, 12, seed=21, kernel=K)
        assert np.array_equal(renderer.accumulator().view(np.uint32), got.view(np.uint32))


@pytest.mark.parametrize("kernel", [P.KERNEL_WAVEFRONT, P.KERNEL_PERSISTENT])
def test_edge_tile_padding_with_many_samples_per_pixel(kernel):
    """Width and band height not multiples of 8, 128 samples per call: with pixel-major path ids the samples of a padded pixel are
    whole 64-id blocks of padding, and every wave is handed several blocks.  A wave that gets nothing but padding must fetch on,
    not retire (found by the rank-share table: the ray counts of the 132-row shares of a 1080-row frame did not add up to the
    whole frame's).  Checked against the megakernel (one thread per pixel: no ids at all): rays and accumulator bits."""
    v, i = standin_mesh(3)
    o, s = reference_layout_pair(v, i, 3, aspect=196 / 100)
    W, H, spp = 196, 100, 128                     # 24.5 x 12.5 tiles; 41 600 blocks of 64 ids
    for interleave in (None, (4, 2, 1)):          # whole frame; rank 1 of 2's 4-row bands (52 rows: 6.5 tiles)
        ref = P.Renderer(0)
        ref.upload(s)
        ref.render(W, H, spp, seed=3, kernel=P.KERNEL_MEGAKERNEL, interleave=interleave)
        r = P.Renderer(0)
        r.upload(s)
        r.render(W, H, spp, seed=3, kernel=kernel, interleave=interleave)
        assert r.stats().traced_rays == ref.stats().traced_rays
        assert np.array_equal(r.accumulator().view(np.uint32), ref.accumulator().view(np.uint32))
        assert np.all(r.accumulator()[..., 3] == spp)
        ref.close(); r.close()
