"""The multi-device context (cgpt_ctx_create with n_devices > 1, csrc/device/multi_gpu.hip): one context spreads the frame over
its devices in interleaved row bands and gathers the float4 bands with one grouped RCCL exchange.  A one-GPU box cannot host two
RCCL ranks, so the two halves are covered separately, both against a plain one-device context and the oracle:
  * the RCCL path itself with ONE rank (CGPT_CTX_FORCE_COLLECTIVE: tiling, ncclSend/ncclRecv to itself, reorder kernel);
  * the tiling / reorder / statistics / checkpoint logic with 2, 3 and 8 ranks that share the GPU (CGPT_CTX_GATHER_PEER_COPY:
    hipMemcpyPeerAsync in place of RCCL, repeated device ids allowed)."""
import numpy as np
import pytest

import oracle as O
import cpugpupathtracing_amd as P
from cpugpupathtracing_amd import distributed as D
from scenes import MAT_SPEC_DIFFUSE, reference_layout_pair, rmse, standin_mesh

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene_and_single():
    v, i = standin_mesh(3)
    o, s = reference_layout_pair(v, i, 4, aspect=100 / 71, extra_materials=(MAT_SPEC_DIFFUSE,))
    W, H, spp = 100, 71, 6
    o.render(W, H, spp, O.MODE_ADVANCED, O.DEBUG_NONE, O.RNG_PIXEL_PCG, 0x12345678, nthreads=8)
    r = P.Renderer(0)
    r.upload(s)
    r.render(W, H, spp, counters=True)
    single = (r.accumulator().copy(), r.pixels().copy(), r.stats())
    r.close()
    assert np.array_equal(single[0].view(np.uint32), o.accumulator().view(np.uint32))
    return s, o, (W, H, spp), single


def _check_against_single(g, s, dims, single, kernel=P.KERNEL_AUTO):
    W, H, spp = dims
    g.upload(s)
    g.reset_stats()
    g.render(W, H, spp // 2, counters=True, kernel=kernel)
    g.render(W, H, spp - spp // 2, counters=True, kernel=kernel)        # accumulation continues across calls on every device
    acc, px, st = g.accumulator(), g.pixels(), g.stats()
    assert acc.shape == (H, W, 4)
    assert np.array_equal(acc.view(np.uint32), single[0].view(np.uint32))
    assert np.array_equal(px, single[1])
    s1 = single[2]
    assert (st.traced_rays, st.inner_steps, st.tri_tests, st.bvh_depth_sum, st.closest_hits, st.num_accumulated) == \
           (s1.traced_rays, s1.inner_steps, s1.tri_tests, s1.bvh_depth_sum, s1.closest_hits, spp)
    assert abs(st.total_energy_received - s1.total_energy_received) < 1e-6 * max(1.0, s1.total_energy_received)


def test_rccl_path_with_one_rank(scene_and_single):
    s, o, dims, single = scene_and_single
    g = P.Renderer([0], flags=P.CTX_FORCE_COLLECTIVE)
    assert g.is_group
    _check_against_single(g, s, dims, single)
    # reset and a debug view: data.pixels is gathered, not re-packed
    g.reset_accumulator()
    assert not g.accumulator().any()
    st = P.Settings(debug_render_mode=P.DEBUG_BVH_DEPTH)
    o.render(dims[0], dims[1], 1, O.MODE_ADVANCED, O.DEBUG_BVH_DEPTH, O.RNG_PIXEL_PCG, 3, nthreads=2)
    g.render(dims[0], dims[1], 1, seed=3, settings=st)
    assert np.array_equal(g.pixels(), o.pixels())
    g.close()


@pytest.mark.parametrize("ranks,kernel", [(2, P.KERNEL_AUTO), (3, P.KERNEL_WAVEFRONT), (3, P.KERNEL_PERSISTENT), (8, P.KERNEL_MEGAKERNEL)])
def test_tiling_over_ranks_sharing_the_gpu(scene_and_single, ranks, kernel):
    s, o, dims, single = scene_and_single
    g = P.Renderer([0] * ranks, flags=P.CTX_GATHER_PEER_COPY)
    _check_against_single(g, s, dims, single, kernel=kernel)
    g.close()


def test_checkpoint_resume_and_tuning_on_a_group(scene_and_single):
    s, o, dims, single = scene_and_single
    W, H, spp = dims
    a = P.Renderer([0, 0, 0], flags=P.CTX_GATHER_PEER_COPY)
    a.upload(s)
    a.set_tuning(band_rows=8, pools=1)
    a.render(W, H, 4)
    saved = a.accumulator().copy()
    a.close()
    b = P.Renderer([0, 0], flags=P.CTX_GATHER_PEER_COPY)          # a different device count: the saved frame is scattered over its own bands
    b.upload(s)
    b.load_accumulator(saved, 4, W, H)
    assert np.array_equal(b.accumulator().view(np.uint32), saved.view(np.uint32))
    assert np.array_equal(b.pixels(), D.pack_pixels(saved, 4))
    b.render(W, H, spp - 4)
    assert np.array_equal(b.accumulator().view(np.uint32), single[0].view(np.uint32))
    with pytest.raises(P.DeviceError, match="tiles the image itself"):
        b.render(W, H, 1, rows=(0, 8))
    b.close()


def test_band_rows_changed_between_two_accumulating_renders_keeps_the_sums(scene_and_single):
    """cgpt_set_tuning never changes results: re-tiling a frame that is being accumulated moves the sums to the new bands"""
    s, o, dims, single = scene_and_single
    W, H, spp = dims
    g = P.Renderer([0, 0, 0], flags=P.CTX_GATHER_PEER_COPY)
    g.upload(s)
    g.set_tuning(band_rows=2)
    g.render(W, H, 2)
    g.set_tuning(band_rows=16)                                       # mid-accumulation: gather under 2-row bands, scatter under 16-row bands
    assert g.stats().num_accumulated == 2
    g.render(W, H, 1)
    g.set_tuning(band_rows=16)                                       # unchanged: nothing moves
    g.set_tuning(band_rows=1)
    g.render(W, H, spp - 3)
    assert np.array_equal(g.accumulator().view(np.uint32), single[0].view(np.uint32))
    assert np.array_equal(g.pixels(), single[1])
    st = g.stats()
    assert st.num_accumulated == spp and st.n_devices == 3 and st.rccl_ranks == 0
    g.close()


def test_group_stats_report_the_exchange_and_every_device(scene_and_single):
    s, o, dims, single = scene_and_single
    W, H, spp = dims
    g = P.Renderer([0], flags=P.CTX_FORCE_COLLECTIVE)
    g.upload(s)
    g.reset_stats()
    g.render(W, H, spp, kernel=P.KERNEL_PERSISTENT)
    g.accumulator(); g.accumulator()                                 # the second read finds the frame gathered already
    st = g.stats()
    assert st.n_devices == 1 and st.rccl_ranks == 1 and st.gathers == 1 and st.gather_ms > 0
    assert st.last_kernel == P.KERNEL_PERSISTENT and st.device_ms[0] == st.kernel_ms > 0 and st.device_ms[1] == 0
    g.render(W, H, 1, kernel=P.KERNEL_WAVEFRONT)
    g.pixels()
    st = g.stats()
    assert st.gathers == 2 and st.last_kernel == P.KERNEL_WAVEFRONT and st.dominant_round0_launches >= 1 and 0 < st.dominant_round0_ms <= st.dominant_ms
    g.reset_stats()
    assert g.stats().gathers == 0 and g.stats().gather_ms == 0
    # a call forwarded to the first device leaves its message in the group context
    with pytest.raises(P.DeviceError, match="device 0: .*cgpt_measure_issue_rate"):
        g.measure_issue_rate(kind=99)
    g.close()
    one = P.Renderer(0)
    one.upload(s)
    one.render(W, H, 1)
    st = one.stats()
    assert st.n_devices == 1 and st.rccl_ranks == 0 and st.gathers == 0 and st.last_kernel in (P.KERNEL_MEGAKERNEL, P.KERNEL_PERSISTENT, P.KERNEL_WAVEFRONT)
    one.close()


def test_group_creation_errors():
    with pytest.raises(P.DeviceError, match="listed twice"):
        P.Renderer([0, 0])                                          # RCCL wants one rank per GPU
    with pytest.raises(P.DeviceError, match="outside"):
        P.Renderer(list(range(9)))
    with pytest.raises(P.DeviceError):
        P.Renderer([0, 99], flags=P.CTX_GATHER_PEER_COPY)


def test_random_groups_agree_with_one_device():
    """random frame sizes, rank counts, band heights, kernels, sample counts and settings: the group's gathered frame, packed pixels and
    ray count equal a one-device context's (CGPT_FUZZ_CASES / CGPT_FUZZ_SEED widen the run)"""
    import os
    cases = max(4, int(os.environ.get("CGPT_FUZZ_CASES", "16")) // 2)
    rng = np.random.default_rng(int(os.environ.get("CGPT_FUZZ_SEED", "1")) + 1000)
    v, i = standin_mesh(2)
    for case in range(cases):
        W, H = int(rng.integers(17, 300)), int(rng.integers(9, 200))
        spp = int(rng.choice([1, 2, 5, 9, 33, 70]))
        mat = int(rng.choice([1, 3, 4]))
        mode = int(rng.choice([P.MODE_ADVANCED, P.MODE_ADVANCED, P.MODE_COMPARISON]))
        st = P.Settings(max_ray_depth=int(rng.choice([1, 5])), render_mode=mode, debug_render_mode=int(rng.choice([P.DEBUG_NONE, P.DEBUG_NONE, P.DEBUG_RAY_DEPTH])))
        kernel = int(rng.choice([P.KERNEL_AUTO, P.KERNEL_PERSISTENT, P.KERNEL_MEGAKERNEL, P.KERNEL_WAVEFRONT]))
        ranks, band_rows = int(rng.integers(2, 7)), int(rng.choice([1, 4, 8, 16]))
        seed = int(rng.integers(0, 2 ** 31))
        o, s = reference_layout_pair(v, i, mat, aspect=W / H, extra_materials=(MAT_SPEC_DIFFUSE,), settings=st)
        s.set_settings(st)
        one = P.Renderer(0)
        one.upload(s)
        one.render(W, H, spp, seed=seed, kernel=kernel)
        g = P.Renderer([0] * ranks, flags=P.CTX_GATHER_PEER_COPY)
        g.upload(s)
        g.set_tuning(band_rows=band_rows)
        g.render(W, H, spp, seed=seed, kernel=kernel)
        what = f"case {case}: {W}x{H} spp {spp} mat {mat} mode {mode} debug {st.debug_render_mode} kernel {kernel} ranks {ranks} band_rows {band_rows}"
        assert np.array_equal(g.accumulator().view(np.uint32), one.accumulator().view(np.uint32)), what
        assert np.array_equal(g.pixels(), one.pixels()), what
        assert g.stats().traced_rays == one.stats().traced_rays, what
        g.close(); one.close()
