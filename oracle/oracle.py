"""ctypes binding of the CPU oracle (oracle/libpt_oracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (cpugpupathtracing_amd/) never does.  PARITY UNPINNED (no reference fixture exists and the
reference cannot be built here): oracle/pt_oracle.h says what anchors the restatement instead.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpt_oracle.so")

BUILD_NAIVE, BUILD_SAH_INTERVALS, BUILD_SAH_PRIMITIVES = 0, 1, 2
MODE_COMPARISON, MODE_BRUTE_FORCE, MODE_ADVANCED = 0, 1, 2
DEBUG_NONE, DEBUG_RAY_DEPTH, DEBUG_BVH_DEPTH = 0, 1, 2
RNG_REFERENCE_XORSHIFT, RNG_PIXEL_PCG = 0, 1


class Stats(C.Structure):
    _fields_ = [("traced_rays", C.c_uint64), ("inner_steps", C.c_uint64), ("tri_tests", C.c_uint64),
                ("bvh_depth_sum", C.c_uint64), ("closest_hits", C.c_uint64),
                ("total_energy_received", C.c_double)]


class BvhInfo(C.Structure):
    _fields_ = [("num_triangles", C.c_uint32), ("nodes_used", C.c_uint32), ("num_leaves", C.c_uint32),
                ("max_leaf_size", C.c_uint32), ("max_depth", C.c_uint32), ("total_area", C.c_float)]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile).  Building the checker is not using it."""
    src = os.path.join(_HERE, "pt_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(
            os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "pt_oracle.h"))):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libpt_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    f3 = C.POINTER(C.c_float)
    u32p = C.POINTER(C.c_uint32)
    L.orc_scene_new.restype = C.c_void_p
    L.orc_scene_free.argtypes = [C.c_void_p]
    mat_args = [f3, C.c_float, C.c_float, f3, C.c_float, f3, C.c_float, C.c_int]
    L.orc_add_material.argtypes = [C.c_void_p] + mat_args
    L.orc_set_material.argtypes = [C.c_void_p, C.c_int] + mat_args
    L.orc_add_mesh.argtypes = [C.c_void_p, f3, C.c_uint32, u32p, C.c_uint32, C.c_uint32, C.c_int]
    L.orc_add_sphere.argtypes = [C.c_void_p, f3, C.c_float, C.c_uint32]
    L.orc_add_plane.argtypes = [C.c_void_p, f3, f3, C.c_uint32]
    L.orc_add_light.argtypes = [C.c_void_p, C.c_uint32]
    L.orc_set_camera.argtypes = [C.c_void_p, f3, f3, C.c_float, C.c_float]
    L.orc_set_settings.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    L.orc_rebuild_bvh.argtypes = [C.c_void_p, C.c_uint32, C.c_int]
    L.orc_bvh_info_get.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(BvhInfo)]
    L.orc_bvh_export.argtypes = [C.c_void_p, C.c_uint32, u32p, u32p]
    L.orc_reset_accumulator.argtypes = [C.c_void_p]
    L.orc_render.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int,
                             C.c_uint32, C.c_int, C.c_uint32, C.c_uint32]
    L.orc_accumulator.argtypes = [C.c_void_p]
    L.orc_accumulator.restype = f3
    L.orc_pixels.argtypes = [C.c_void_p]
    L.orc_pixels.restype = u32p
    L.orc_num_accumulated.argtypes = [C.c_void_p]
    L.orc_num_accumulated.restype = C.c_uint32
    L.orc_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
    L.orc_reset_stats.argtypes = [C.c_void_p]
    L.orc_intersect_rays.argtypes = [C.c_void_p, f3, f3, f3, C.c_uint32, f3, u32p, u32p, u32p]
    L.orc_camera_ray.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, f3, f3]
    L.orc_wang_hash.argtypes = [C.c_uint32]
    L.orc_wang_hash.restype = C.c_uint32
    L.orc_pcg_seed.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
    L.orc_pcg_seed.restype = C.c_uint32
    L.orc_pcg_next.argtypes = [u32p]
    L.orc_pcg_next.restype = C.c_uint32
    L.orc_xorshift32.argtypes = [u32p]
    L.orc_xorshift32.restype = C.c_uint32
    L.orc_u32_to_float.argtypes = [C.c_uint32]
    L.orc_u32_to_float.restype = C.c_float
    L.orc_vec4_to_uint.argtypes = [f3]
    L.orc_vec4_to_uint.restype = C.c_uint32
    L.orc_fresnel.argtypes = [C.c_float] * 4
    L.orc_fresnel.restype = C.c_float
    L.orc_reflect.argtypes = [f3, f3, f3]
    L.orc_intersect_triangle.argtypes = [f3, f3, f3, f3, f3, f3]
    L.orc_intersect_sphere.argtypes = [f3, C.c_float, f3, f3, f3]
    L.orc_intersect_aabb.argtypes = [f3, f3, f3, f3, C.c_float]
    L.orc_intersect_aabb.restype = C.c_float
    _lib = L
    return L


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def _u(a):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    return a, a.ctypes.data_as(C.POINTER(C.c_uint32))


class OracleScene:
    """Mirror of the reference's file-static `data` (Main.cpp:200-236) driven through the C oracle."""

    def __init__(self):
        self.L = lib()
        self.h = C.c_void_p(self.L.orc_scene_new())
        self.W = self.H = 0

    def close(self):
        if self.h:
            self.L.orc_scene_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- scene construction -------------------------------------------------------------------
    def add_material(self, albedo=(0, 0, 0), specular=0.0, refractivity=0.0, absorption=(0, 0, 0), ior=1.0,
                     emissive=(0, 0, 0), intensity=0.0, is_light=False):
        _, a = _f(albedo); _, b = _f(absorption); _, e = _f(emissive)
        return self.L.orc_add_material(self.h, a, specular, refractivity, b, ior, e, intensity, int(is_light))

    def set_material(self, index, albedo=(0, 0, 0), specular=0.0, refractivity=0.0, absorption=(0, 0, 0), ior=1.0,
                     emissive=(0, 0, 0), intensity=0.0, is_light=False):
        _, a = _f(albedo); _, b = _f(absorption); _, e = _f(emissive)
        rc = self.L.orc_set_material(self.h, index, a, specular, refractivity, b, ior, e, intensity, int(is_light))
        assert rc == 0

    def add_mesh(self, vertices, indices, mat_index, build_option=BUILD_SAH_INTERVALS):
        v, vp = _f(vertices)
        i, ip = _u(indices)
        assert v.ndim == 2 and v.shape[1] == 6
        rc = self.L.orc_add_mesh(self.h, vp, v.shape[0], ip, i.size, mat_index, build_option)
        if rc < 0:
            raise ValueError(f"orc_add_mesh failed rc={rc}")
        return rc

    def add_sphere(self, center, radius, mat_index):
        _, c = _f(center)
        return self.L.orc_add_sphere(self.h, c, radius, mat_index)

    def add_plane(self, normal, point, mat_index):
        _, n = _f(normal); _, p = _f(point)
        return self.L.orc_add_plane(self.h, n, p, mat_index)

    def add_light(self, obj_index):
        rc = self.L.orc_add_light(self.h, obj_index)
        if rc != 0:
            raise ValueError(f"orc_add_light rc={rc}")

    def set_camera(self, pos, view_dir, fov_deg, aspect):
        _, p = _f(pos); _, d = _f(view_dir)
        self.L.orc_set_camera(self.h, p, d, fov_deg, aspect)

    def set_settings(self, max_ray_depth=5, nee=True, cosine=True, rr=True):
        self.L.orc_set_settings(self.h, max_ray_depth, int(nee), int(cosine), int(rr))

    def rebuild_bvh(self, obj_index, build_option):
        assert self.L.orc_rebuild_bvh(self.h, obj_index, build_option) == 0

    # --- BVH inspection -----------------------------------------------------------------------
    def bvh_info(self, obj_index):
        info = BvhInfo()
        assert self.L.orc_bvh_info_get(self.h, obj_index, C.byref(info)) == 0
        return info

    def bvh_export(self, obj_index):
        info = self.bvh_info(obj_index)
        nodes = np.zeros((info.nodes_used, 8), dtype=np.uint32)
        tri = np.zeros(info.num_triangles, dtype=np.uint32)
        assert self.L.orc_bvh_export(self.h, obj_index, nodes.ctypes.data_as(C.POINTER(C.c_uint32)),
                                     tri.ctypes.data_as(C.POINTER(C.c_uint32))) == 0
        return nodes, tri

    # --- rendering ----------------------------------------------------------------------------
    def reset_accumulator(self):
        self.L.orc_reset_accumulator(self.h)

    def render(self, W, H, n_frames=1, render_mode=MODE_ADVANCED, debug_mode=DEBUG_NONE,
               rng_mode=RNG_PIXEL_PCG, seed=0x12345678, nthreads=1, rows=None):
        r0, r1 = (0, H) if rows is None else rows
        rc = self.L.orc_render(self.h, W, H, n_frames, render_mode, debug_mode, rng_mode, seed, nthreads, r0, r1)
        if rc != 0:
            raise ValueError(f"orc_render rc={rc}")
        self.W, self.H = W, H

    def accumulator(self):
        p = self.L.orc_accumulator(self.h)
        return np.ctypeslib.as_array(p, shape=(self.H, self.W, 4)).copy()

    def pixels(self):
        p = self.L.orc_pixels(self.h)
        return np.ctypeslib.as_array(p, shape=(self.H, self.W)).copy()

    def num_accumulated(self):
        return self.L.orc_num_accumulated(self.h)

    def stats(self):
        s = Stats()
        self.L.orc_get_stats(self.h, C.byref(s))
        return s

    def reset_stats(self):
        self.L.orc_reset_stats(self.h)

    def intersect_rays(self, origins, dirs, tmax=None):
        o, op = _f(origins); d, dp = _f(dirs)
        n = o.shape[0]
        if tmax is None:
            tmax = np.full(n, 1e34, dtype=np.float32)
        t, tp = _f(tmax)
        out_t = np.zeros(n, np.float32); out_obj = np.zeros(n, np.uint32)
        out_tri = np.zeros(n, np.uint32); out_depth = np.zeros(n, np.uint32)
        self.L.orc_intersect_rays(self.h, op, dp, tp, n, out_t.ctypes.data_as(C.POINTER(C.c_float)),
                                  out_obj.ctypes.data_as(C.POINTER(C.c_uint32)),
                                  out_tri.ctypes.data_as(C.POINTER(C.c_uint32)),
                                  out_depth.ctypes.data_as(C.POINTER(C.c_uint32)))
        return out_t, out_obj, out_tri, out_depth

    def camera_rays(self, W, H):
        o = np.zeros((H, W, 3), np.float32); d = np.zeros((H, W, 3), np.float32)
        oo = (C.c_float * 3)(); dd = (C.c_float * 3)()
        for y in range(H):
            for x in range(W):
                self.L.orc_camera_ray(self.h, x, y, W, H, oo, dd)
                o[y, x] = oo[:]; d[y, x] = dd[:]
        return o, d


def load_gltf_reference_semantics(path: str):
    """Python restatement of GLTFLoader::Load (ref: Source/GLTFLoader.cpp:19-89).

    Last primitive of the last mesh wins (:34-43), POSITION + NORMAL only (:62-82), u16 indices widened
    (:52-60), accessor.byteOffset + bufferView.byteOffset (:9-17), byteStride ignored.  Returns
    (vertices[n,6] float32, indices[m] uint32).  Raises on a missing buffer file (the reference
    null-derefs there).
    """
    with open(path, "r") as f:
        doc = json.load(f)
    base = os.path.dirname(path)
    bufs = []
    for b in doc["buffers"]:
        with open(os.path.join(base, b["uri"]), "rb") as f:
            bufs.append(f.read())
    verts = None
    idx = None
    for mesh in doc["meshes"]:
        for prim in mesh["primitives"]:
            def ptr(acc_i):
                acc = doc["accessors"][acc_i]
                bv = doc["bufferViews"][acc["bufferView"]]
                return acc, bufs[bv["buffer"]], bv.get("byteOffset", 0) + acc.get("byteOffset", 0)
            acc, buf, off = ptr(prim["indices"])
            if acc["componentType"] == 5125:
                idx = np.frombuffer(buf, dtype="<u4", count=acc["count"], offset=off).astype(np.uint32)
            elif acc["componentType"] == 5123:
                idx = np.frombuffer(buf, dtype="<u2", count=acc["count"], offset=off).astype(np.uint32)
            else:
                idx = np.zeros(acc["count"], np.uint32)
            attrs = list(prim["attributes"].items())
            nverts = doc["accessors"][attrs[0][1]]["count"]
            verts = np.zeros((nverts, 6), np.float32)
            for name, acc_i in attrs:
                acc, buf, off = ptr(acc_i)
                if name == "POSITION":
                    verts[:acc["count"], 0:3] = np.frombuffer(buf, "<f4", acc["count"] * 3, off).reshape(-1, 3)
                elif name == "NORMAL":
                    verts[:acc["count"], 3:6] = np.frombuffer(buf, "<f4", acc["count"] * 3, off).reshape(-1, 3)
    return verts, idx
