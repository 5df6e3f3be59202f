"""ctypes view of libcpugpupt.so: the structs and prototypes of include/cpugpupt_abi.h and include/cpugpupt_host.h.

The library is the product; if it is missing or a symbol is absent this module raises -- there is no Python or CPU
fallback for the render path.
"""
from __future__ import annotations

import ctypes as C
import os

from ._paths import LIB_PATH

ABI_VERSION = 2
CGPT_OK, CGPT_ERR_INVALID, CGPT_ERR_HIP, CGPT_ERR_NO_SCENE, CGPT_ERR_UNSUPPORTED, CGPT_ERR_NO_DEVICE = range(6)
OBJECT_MESH, OBJECT_SPHERE, OBJECT_PLANE = 0, 1, 2
MODE_COMPARISON, MODE_BRUTE_FORCE, MODE_ADVANCED = 0, 1, 2
DEBUG_NONE, DEBUG_RAY_DEPTH, DEBUG_BVH_DEPTH = 0, 1, 2
KERNEL_AUTO, KERNEL_MEGAKERNEL, KERNEL_WAVEFRONT, KERNEL_PERSISTENT = 0, 1, 2, 3
RENDER_COUNTERS = 1
CTX_FORCE_COLLECTIVE, CTX_GATHER_PEER_COPY = 1, 2
BUILD_NAIVE, BUILD_SAH_INTERVALS, BUILD_SAH_PRIMITIVES = 0, 1, 2

f3 = C.c_float * 3


class Vertex(C.Structure):
    _fields_ = [("pos", f3), ("normal", f3)]


class Triangle(C.Structure):
    _fields_ = [("v0", Vertex), ("v1", Vertex), ("v2", Vertex)]


class BvhNode(C.Structure):
    _fields_ = [("aabb_min", f3), ("left_first", C.c_uint32), ("aabb_max", f3), ("prim_count", C.c_uint32)]


class Material(C.Structure):
    _fields_ = [("albedo", f3), ("specular", C.c_float), ("refractivity", C.c_float), ("absorption", f3),
                ("ior", C.c_float), ("emissive", f3), ("intensity", C.c_float), ("is_light", C.c_uint32)]


class Object(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("mat_index", C.c_uint32), ("node_offset", C.c_uint32), ("node_count", C.c_uint32),
                ("tri_offset", C.c_uint32), ("tri_count", C.c_uint32), ("max_depth", C.c_uint32), ("total_area", C.c_float),
                ("sphere_center", f3), ("sphere_radius", C.c_float), ("plane_normal", f3), ("plane_point", f3)]


class SceneDesc(C.Structure):
    _fields_ = [("objects", C.POINTER(Object)), ("n_objects", C.c_uint32),
                ("nodes", C.POINTER(BvhNode)), ("n_nodes", C.c_uint32),
                ("triangles", C.POINTER(Triangle)), ("n_triangles", C.c_uint32),
                ("tri_indices", C.POINTER(C.c_uint32)),
                ("materials", C.POINTER(Material)), ("n_materials", C.c_uint32),
                ("light_indices", C.POINTER(C.c_uint32)), ("n_lights", C.c_uint32)]


class Camera(C.Structure):
    _fields_ = [("pos", f3), ("top_left", f3), ("top_right", f3), ("bottom_left", f3)]


class Settings(C.Structure):
    _fields_ = [("max_ray_depth", C.c_int32), ("next_event_estimation_enabled", C.c_uint32),
                ("cosine_weighted_diffuse_reflection_enabled", C.c_uint32), ("russian_roulette_enabled", C.c_uint32),
                ("render_mode", C.c_uint32), ("debug_render_mode", C.c_uint32)]


class RenderParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("row_begin", C.c_uint32), ("row_end", C.c_uint32),
                ("first_sample", C.c_uint32), ("n_samples", C.c_uint32), ("seed", C.c_uint32), ("kernel", C.c_uint32),
                ("flags", C.c_uint32), ("interleave_rows", C.c_uint32), ("interleave_count", C.c_uint32),
                ("interleave_index", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("traced_rays", C.c_uint64), ("inner_steps", C.c_uint64), ("tri_tests", C.c_uint64),
                ("bvh_depth_sum", C.c_uint64), ("closest_hits", C.c_uint64), ("total_energy_received", C.c_double),
                ("num_accumulated", C.c_uint32), ("kernel_launches", C.c_uint32), ("kernel_ms", C.c_double),
                ("dominant_launches", C.c_uint32), ("dominant_waves_per_simd", C.c_uint32), ("dominant_ms", C.c_double),
                ("gather_ms", C.c_double), ("gathers", C.c_uint32), ("n_devices", C.c_uint32), ("rccl_ranks", C.c_uint32),
                ("last_kernel", C.c_uint32), ("device_ms", C.c_double * 8),
                ("dominant_round0_ms", C.c_double), ("dominant_round0_launches", C.c_uint32), ("reserved_", C.c_uint32)]


class BvhInfo(C.Structure):
    _fields_ = [("num_triangles", C.c_uint32), ("nodes_used", C.c_uint32), ("num_leaves", C.c_uint32),
                ("max_leaf_size", C.c_uint32), ("max_depth", C.c_uint32), ("total_area", C.c_float)]


_vp = C.c_void_p
_fp = C.POINTER(C.c_float)
_up = C.POINTER(C.c_uint32)

# name -> (restype, argtypes).  Every symbol declared in include/*.h is listed; tests check the list against the headers.
PROTOTYPES = {
    # cpugpupt_abi.h
    "cgpt_abi_version": (C.c_uint32, []),
    "cgpt_ctx_create": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.c_uint32, C.POINTER(_vp)]),
    "cgpt_ctx_destroy": (C.c_int, [_vp]),
    "cgpt_last_error": (C.c_char_p, [_vp]),
    "cgpt_set_stream": (C.c_int, [_vp, _vp]),
    "cgpt_scene_upload": (C.c_int, [_vp, C.POINTER(SceneDesc)]),
    "cgpt_scene_update_materials": (C.c_int, [_vp, C.POINTER(Material), C.c_uint32]),
    "cgpt_camera_from_view": (C.c_int, [_fp, _fp, C.c_float, C.c_float, C.POINTER(Camera)]),
    "cgpt_render": (C.c_int, [_vp, C.POINTER(Camera), C.POINTER(Settings), C.POINTER(RenderParams)]),
    "cgpt_reset_accumulator": (C.c_int, [_vp]),
    "cgpt_read_accumulator": (C.c_int, [_vp, _fp, C.c_size_t]),
    "cgpt_read_pixels": (C.c_int, [_vp, _up, C.c_size_t]),
    "cgpt_accumulator_device_ptr": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(C.c_size_t)]),
    "cgpt_pixels_device_ptr": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(C.c_size_t)]),
    "cgpt_get_stats": (C.c_int, [_vp, C.POINTER(Stats)]),
    "cgpt_reset_stats": (C.c_int, [_vp]),
    "cgpt_intersect_rays": (C.c_int, [_vp, _fp, _fp, _fp, C.c_uint32, _fp, _up, _up, _up]),
    "cgpt_bvh_build": (C.c_int, [_vp, C.POINTER(Triangle), C.c_uint32, C.POINTER(BvhNode), _up, _up, _up, _fp]),
    "cgpt_bvh_build_ex": (C.c_int, [_vp, C.POINTER(Triangle), C.c_uint32, C.c_uint32, _up, C.POINTER(BvhNode), _up, _up, _up, _fp]),
    "cgpt_synchronize": (C.c_int, [_vp]),
    "cgpt_write_accumulator": (C.c_int, [_vp, C.POINTER(RenderParams), _fp, C.c_size_t, C.c_uint32]),
    "cgpt_set_tuning": (C.c_int, [_vp, C.c_char_p, C.c_uint32]),
    "cgpt_measure_issue_rate": (C.c_int, [_vp, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    # cpugpupt_host.h
    "cgpth_last_error": (C.c_char_p, []),
    "cgpth_mesh_load_gltf": (_vp, [C.c_char_p]),
    "cgpth_mesh_from_arrays": (_vp, [C.POINTER(Vertex), C.c_uint32, _up, C.c_uint32]),
    "cgpth_mesh_dragon_standin": (_vp, [C.c_uint32]),
    "cgpth_mesh_bumpy_icosphere": (_vp, [C.c_uint32, _fp, _fp, C.c_float]),
    "cgpth_mesh_save_gltf": (C.c_int, [_vp, C.c_char_p]),
    "cgpth_mesh_num_vertices": (C.c_uint32, [_vp]),
    "cgpth_mesh_num_indices": (C.c_uint32, [_vp]),
    "cgpth_mesh_vertices": (C.POINTER(Vertex), [_vp]),
    "cgpth_mesh_indices": (_up, [_vp]),
    "cgpth_mesh_free": (None, [_vp]),
    "cgpth_scene_new": (_vp, []),
    "cgpth_scene_free": (None, [_vp]),
    "cgpth_scene_reference_layout": (_vp, [_vp, C.c_uint32, C.c_float, C.c_int]),
    "cgpth_scene_add_material": (C.c_int, [_vp, C.POINTER(Material)]),
    "cgpth_scene_set_material": (C.c_int, [_vp, C.c_uint32, C.POINTER(Material)]),
    "cgpth_scene_add_mesh": (C.c_int, [_vp, _vp, C.c_uint32, C.c_int]),
    "cgpth_scene_add_mesh_device_built": (C.c_int, [_vp, _vp, C.c_uint32, _vp]),
    "cgpth_scene_add_mesh_device_built_ex": (C.c_int, [_vp, _vp, C.c_uint32, _vp, C.c_int]),
    "cgpth_scene_rebuild_bvh_device": (C.c_int, [_vp, C.c_uint32, C.c_int, _vp]),
    "cgpth_scene_add_sphere": (C.c_int, [_vp, _fp, C.c_float, C.c_uint32]),
    "cgpth_scene_add_plane": (C.c_int, [_vp, _fp, _fp, C.c_uint32]),
    "cgpth_scene_add_light": (C.c_int, [_vp, C.c_uint32]),
    "cgpth_scene_set_camera": (C.c_int, [_vp, _fp, _fp, C.c_float, C.c_float]),
    "cgpth_scene_set_settings": (C.c_int, [_vp, C.POINTER(Settings)]),
    "cgpth_scene_rebuild_bvh": (C.c_int, [_vp, C.c_uint32, C.c_int]),
    "cgpth_scene_bvh_info": (C.c_int, [_vp, C.c_uint32, C.POINTER(BvhInfo)]),
    "cgpth_scene_bvh_export": (C.c_int, [_vp, C.c_uint32, C.POINTER(BvhNode), _up]),
    "cgpth_scene_flatten": (C.c_int, [_vp, C.POINTER(SceneDesc)]),
    "cgpth_scene_get_camera": (C.c_int, [_vp, C.POINTER(Camera)]),
    "cgpth_scene_get_settings": (C.c_int, [_vp, C.POINTER(Settings)]),
    "cgpth_write_ppm": (C.c_int, [C.c_char_p, _up, C.c_uint32, C.c_uint32]),
    "cgpth_write_pfm": (C.c_int, [C.c_char_p, _fp, C.c_uint32, C.c_uint32, C.c_uint32]),
    "cgpth_write_accumulator": (C.c_int, [C.c_char_p, _fp, C.c_uint32, C.c_uint32, C.c_uint32]),
    "cgpth_read_accumulator": (C.c_int, [C.c_char_p, _fp, _up, C.c_uint32, C.c_uint32]),
    "cgpth_fast_div": (C.c_uint32, [C.c_uint32, C.c_uint32]),
}

_lib = None


class NativeLibraryError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Loads libcpugpupt.so (built in-tree by cpugpupathtracing_amd.build).  Fails loudly when absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} is missing: run `python -m cpugpupathtracing_amd.build` (hipcc, gfx950). "
            "There is no CPU or Python fallback for the render path.")
    L = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in PROTOTYPES.items():
        try:
            fn = getattr(L, name)
        except AttributeError as e:
            raise NativeLibraryError(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.restype = restype
        fn.argtypes = argtypes
    if L.cgpt_abi_version() != ABI_VERSION:
        raise NativeLibraryError(f"ABI version mismatch: library reports {L.cgpt_abi_version()}, binding expects {ABI_VERSION}")
    _lib = L
    return L
