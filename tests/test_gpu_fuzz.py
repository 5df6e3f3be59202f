"""Randomised cross-check of the three render paths (and, for small frames, of the oracle) through the C ABI: random frame sizes (not
multiples of the tile), samples per call, sample offsets, bands / interleaved bands, settings, render and debug modes, mesh materials
and tuning knobs.  Every case: accumulators and packed pixels bit-identical across kernels, traced_rays equal; against the oracle the
megakernel is bit-identical on scenes without glass and within 1e-4 RMSE with it (expf ULPs).
CGPT_FUZZ_CASES / CGPT_FUZZ_SEED widen the run (default: 24 cases, seed 1; a failing case prints its parameters)."""
import os
import time

import numpy as np
import pytest

import oracle as O
import cpugpupathtracing_amd as P
from scenes import MAT_SPEC_DIFFUSE, reference_layout_pair, standin_mesh

pytestmark = pytest.mark.gpu


def test_random_cases_agree_across_kernels_and_with_the_oracle():
    cases = int(os.environ.get("CGPT_FUZZ_CASES", "24"))
    rng = np.random.default_rng(int(os.environ.get("CGPT_FUZZ_SEED", "1")))
    meshes = {lv: standin_mesh(lv) for lv in (1, 2, 3, 4)}
    t_start = time.time()
    n_oracle = 0
    for case in range(cases):
        lv = int(rng.choice([1, 2, 3, 4]))
        mat = int(rng.choice([0, 1, 3, 4]))
        big = rng.random() < 0.15                     # now and then a frame with many 64-id blocks per wave of the persistent grids
        W, H = (int(rng.integers(300, 1700)), int(rng.integers(200, 1000))) if big else (int(rng.integers(9, 420)), int(rng.integers(5, 300)))
        spp = int(rng.choice([1, 2, 3, 5, 8, 13, 33, 64, 65, 130, 256]))
        budget = 60_000_000 if big else 6_000_000
        if W * H * spp > budget:
            spp = max(1, budget // (W * H))
        first = int(rng.choice([0, 0, 1, 7, 300]))
        seed = int(rng.integers(0, 2 ** 31))
        mode = int(rng.choice([P.MODE_ADVANCED] * 3 + [P.MODE_BRUTE_FORCE, P.MODE_COMPARISON]))
        debug = int(rng.choice([P.DEBUG_NONE] * 4 + [P.DEBUG_RAY_DEPTH, P.DEBUG_BVH_DEPTH]))
        st = P.Settings(max_ray_depth=int(rng.choice([0, 1, 3, 5, 5, 7])), next_event_estimation_enabled=bool(rng.random() < 0.8),
                        cosine_weighted_diffuse_reflection_enabled=bool(rng.random() < 0.7), russian_roulette_enabled=bool(rng.random() < 0.7),
                        render_mode=mode, debug_render_mode=debug)
        band = int(rng.choice([0, 0, 1, 2]))
        rows = interleave = None
        if band == 1:
            a = int(rng.integers(0, H)); rows = (a, int(rng.integers(a + 1, H + 1)))
        elif band == 2:
            world = int(rng.integers(2, 5)); interleave = (int(rng.choice([1, 3, 4, 8])), world, int(rng.integers(0, world)))
            if interleave[2] * interleave[0] >= H:
                interleave = None
        knobs_wf = [{}, {"batch": int(rng.integers(1, 9))}, {"path_order": int(rng.integers(0, 3))}, {"retire_misses": 0}, {"pools": 1}, {"first_lean": 0}, {"bands": int(rng.integers(2, 33)), "bands_min_paths": 0}][int(rng.integers(0, 7))]
        knobs_pt = [{}, {"pt_max_paths_mi": 2 if big else 1}, {"pt_path_order": int(rng.integers(0, 3))}, {"pt_streams": 1}, {"pt_refill": 1, "pt_fine_rounds": 0},
                    {"pt_tail_samples": 0}, {"pt_tail_samples": 4096, "pt_tail_lanes": 64}][int(rng.integers(0, 7))]
        o, s = reference_layout_pair(*meshes[lv], mat, aspect=W / H, extra_materials=(MAT_SPEC_DIFFUSE,), settings=st)
        s.set_settings(st)
        desc = f"case {case}: level {lv} mat {mat} {W}x{H} spp {spp} first {first} mode {mode} debug {debug} depth {st.max_ray_depth} nee {st.next_event_estimation_enabled} cos {st.cosine_weighted_diffuse_reflection_enabled} rr {st.russian_roulette_enabled} rows {rows} il {interleave} wf {knobs_wf} pt {knobs_pt}"
        results = {}
        kernels = [("mega", P.KERNEL_MEGAKERNEL, {}), ("pers", P.KERNEL_PERSISTENT, knobs_pt), ("wave", P.KERNEL_WAVEFRONT, knobs_wf)]
        for name, k, knobs in kernels:
            r = P.Renderer(0)
            r.upload(s)
            if knobs:
                r.set_tuning(**knobs)
            if first:
                r.render(W, H, first, seed=seed, kernel=P.KERNEL_MEGAKERNEL, rows=rows, interleave=interleave)   # same history for all
            r.render(W, H, spp, seed=seed, kernel=k, rows=rows, interleave=interleave)
            results[name] = (r.accumulator().copy(), r.stats().traced_rays, r.pixels().copy() if hasattr(r, "pixels") else None)
            r.close()
        ref = results["mega"]
        for name, (acc, rays, px) in results.items():
            if not np.array_equal(acc.view(np.uint32), ref[0].view(np.uint32)) or rays != ref[1] or (px is not None and not np.array_equal(px, ref[2])):
                bad = np.argwhere((acc.view(np.uint32) != ref[0].view(np.uint32)).any(axis=-1))
                print("MISMATCH", name, desc, "rays", rays, ref[1], "first bad pixels", bad[:5].tolist(), flush=True)
                raise AssertionError("kernels disagree: " + desc)
        if W * H * (spp + first) <= 150_000 and rows is None and interleave is None:
            o.render(W, H, spp + first, {P.MODE_ADVANCED: O.MODE_ADVANCED, P.MODE_BRUTE_FORCE: O.MODE_BRUTE_FORCE, P.MODE_COMPARISON: O.MODE_COMPARISON}[mode],
                     {P.DEBUG_NONE: O.DEBUG_NONE, P.DEBUG_RAY_DEPTH: O.DEBUG_RAY_DEPTH, P.DEBUG_BVH_DEPTH: O.DEBUG_BVH_DEPTH}[debug], O.RNG_PIXEL_PCG, seed, nthreads=8)
            want = o.accumulator()
            n_oracle += 1
            if debug == P.DEBUG_NONE:
                if mat != 3:
                    ok = np.array_equal(want.view(np.uint32), ref[0].view(np.uint32))
                else:
                    d = want.astype(np.float64) - ref[0].astype(np.float64)
                    ok = float(np.sqrt(np.mean(d[..., :3] ** 2))) / (spp + first) < 1e-4
                if not ok or o.stats().traced_rays != ref[1]:
                    print("ORACLE MISMATCH", desc, o.stats().traced_rays, ref[1], flush=True)
                    raise AssertionError("oracle disagrees: " + desc)
        if case % 10 == 9:
            print(f"{case + 1} cases ok ({n_oracle} against the oracle), {time.time() - t_start:.0f} s", flush=True)
    print(f"all {cases} cases ok ({n_oracle} against the oracle), {time.time() - t_start:.0f} s")
    assert n_oracle > 0 or cases < 10
