"""Analytic known-answer tests of the oracle, straight from the cited reference formulas."""
import ctypes as C

import numpy as np

import oracle as O

L = O.lib()
f3 = C.c_float * 3


def test_xorshift32_first_values():
    # ref: Random.h:15-21 with s_seed = 0x12345678: x ^= x<<13; x ^= x>>17; x ^= x<<5
    def ref(x):
        x ^= (x << 13) & 0xFFFFFFFF; x ^= x >> 17; x ^= (x << 5) & 0xFFFFFFFF
        return x
    s = C.c_uint32(0x12345678)
    x = 0x12345678
    for _ in range(5):
        x = ref(x)
        assert L.orc_xorshift32(C.byref(s)) == x


def test_wang_hash_matches_formula():
    def ref(s):
        s = (s ^ 61) ^ (s >> 16); s = (s * 9) & 0xFFFFFFFF; s ^= s >> 4
        s = (s * 0x27d4eb2d) & 0xFFFFFFFF; s ^= s >> 15
        return s
    for v in (0, 1, 0x12345678, 0xFFFFFFFF):
        assert L.orc_wang_hash(v) == ref(v)


def test_random_float_can_be_exactly_one():
    # SURVEY A-17: u32 -> float rounding makes values >= 2^32-128 map to 1.0f
    assert L.orc_u32_to_float(0xFFFFFFFF) == 1.0
    assert L.orc_u32_to_float(0) == 0.0
    assert abs(L.orc_u32_to_float(0x80000000) - 0.5) < 1e-7


def test_pcg_streams_differ_by_pixel_and_sample():
    seeds = {L.orc_pcg_seed(p, s, 0x12345678) for p in range(64) for s in range(8)}
    assert len(seeds) == 64 * 8
    st = C.c_uint32(L.orc_pcg_seed(3, 1, 7))
    a = [L.orc_pcg_next(C.byref(st)) for _ in range(4)]
    st2 = C.c_uint32(L.orc_pcg_seed(3, 1, 7))
    assert a == [L.orc_pcg_next(C.byref(st2)) for _ in range(4)]


def test_vec4_to_uint_packing():
    # ref: MathLib.h:144-152: R | G<<8 | B<<16 | 0xFF<<24, truncation, min(1,.)
    assert L.orc_vec4_to_uint((C.c_float * 4)(1.0, 0.5, 0.0, 1.0)) == 0xFF007FFF
    assert L.orc_vec4_to_uint((C.c_float * 4)(2.0, 0.999, 0.25, 0.0)) == (0xFF << 24) | (63 << 16) | (254 << 8) | 255
    assert L.orc_vec4_to_uint((C.c_float * 4)(-1.0, 0.0, 0.0, 0.0)) == 0xFF000000   # A-15: negative clamped


def test_fresnel_normal_incidence():
    # air -> glass 1.517 at normal incidence: ((1-1.517)/(1+1.517))^2 = 0.04219
    fr = L.orc_fresnel(-1.0, -1.0, 1.0, 1.517)
    assert abs(fr - ((1 - 1.517) / (1 + 1.517)) ** 2) < 1e-6


def test_reflect():
    out = f3()
    L.orc_reflect(f3(1, -1, 0), f3(0, 1, 0), out)
    assert list(out) == [1.0, 1.0, 0.0]


def test_triangle_intersection_and_epsilon():
    t = C.c_float(1e34)
    assert L.orc_intersect_triangle(f3(-1, -1, -5), f3(1, -1, -5), f3(0, 1, -5), f3(0, 0, 0), f3(0, 0, -1), C.byref(t)) == 1
    assert t.value == 5.0
    # double sided (no back-face culling, ref: Primitives.cpp:15-19)
    t = C.c_float(1e34)
    assert L.orc_intersect_triangle(f3(-1, -1, -5), f3(0, 1, -5), f3(1, -1, -5), f3(0, 0, 0), f3(0, 0, -1), C.byref(t)) == 1
    # strict t < ray.t
    t = C.c_float(5.0)
    assert L.orc_intersect_triangle(f3(-1, -1, -5), f3(1, -1, -5), f3(0, 1, -5), f3(0, 0, 0), f3(0, 0, -1), C.byref(t)) == 0
    # absolute determinant epsilon 1e-3 (SURVEY A-9): a 0.02-sized triangle is missed
    t = C.c_float(1e34)
    assert L.orc_intersect_triangle(f3(-.01, -.01, -5), f3(.01, -.01, -5), f3(0, .01, -5), f3(0, 0, 0), f3(0, 0, -1), C.byref(t)) == 0


def test_sphere_intersection_quirks():
    t = C.c_float(1e34)
    assert L.orc_intersect_sphere(f3(0, 0, -5), 1.0, f3(0, 0, 0), f3(0, 0, -1), C.byref(t)) == 1 and t.value == 4.0
    # origin inside, centre behind -> miss (SURVEY A-12)
    t = C.c_float(1e34)
    assert L.orc_intersect_sphere(f3(0, 0, 0.5), 1.0, f3(0, 0, 0), f3(0, 0, -1), C.byref(t)) == 0
    # origin inside, centre ahead -> far root
    t = C.c_float(1e34)
    assert L.orc_intersect_sphere(f3(0, 0, -0.5), 1.0, f3(0, 0, 0), f3(0, 0, -1), C.byref(t)) == 1 and t.value == 1.5


def test_aabb_slab():
    assert L.orc_intersect_aabb(f3(-1, -1, -6), f3(1, 1, -4), f3(0, 0, 0), f3(0, 0, -1), 1e34) == 4.0
    assert L.orc_intersect_aabb(f3(-1, -1, -6), f3(1, 1, -4), f3(0, 0, 0), f3(0, 0, 1), 1e34) == np.float32(1e30)
    assert L.orc_intersect_aabb(f3(-1, -1, -6), f3(1, 1, -4), f3(0, 0, 0), f3(0, 0, -1), 3.0) == np.float32(1e30)   # tmin < ray.t
    # origin inside the box: tmin negative, still a hit
    assert L.orc_intersect_aabb(f3(-1, -1, -1), f3(1, 1, 1), f3(0, 0, 0), f3(0, 0, -1), 1e34) == -1.0


def test_camera_plane_distance_is_fov_in_radians():
    # SURVEY A-13: plane at distance fov-in-radians, half extent (aspect, 1): centre pixel looks down -z
    s = O.OracleScene()
    s.set_camera((0, 0, 8), (0, 0, -1), 60.0, 1.0)
    o = f3(); d = f3()
    L.orc_camera_ray(s.h, 32, 32, 64, 64, o, d)
    assert list(o) == [0.0, 0.0, 8.0] and abs(d[2] + 1.0) < 1e-6
    L.orc_camera_ray(s.h, 0, 0, 64, 64, o, d)
    fov = np.float32(60.0) * np.float32(3.14159265) / np.float32(180.0)
    want = np.array([-1.0, 1.0, -fov]); want /= np.linalg.norm(want)
    assert np.allclose(list(d), want, atol=1e-6)


def test_brute_force_and_advanced_see_the_same_light():
    # a lone emissive sphere seen directly: both integrators return emissive*intensity at the centre pixel
    for mode in (O.MODE_BRUTE_FORCE, O.MODE_ADVANCED):
        s = O.OracleScene()
        s.add_material(emissive=(1.0, 0.5, 0.25), intensity=2.0, is_light=True)
        s.add_light(s.add_sphere((0, 0, -5), 1.0, 0))
        s.set_camera((0, 0, 0), (0, 0, -1), 60.0, 1.0)
        s.render(16, 16, 1, mode, O.DEBUG_NONE, O.RNG_PIXEL_PCG)
        assert list(s.accumulator()[8, 8]) == [2.0, 1.0, 0.5, 1.0]
        assert s.stats().traced_rays == 256
