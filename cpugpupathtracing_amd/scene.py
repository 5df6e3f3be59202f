"""Python handles on the host mirror (include/cpugpupt_host.h): Mesh, Scene, materials, camera, settings.

Names follow the reference (ref: Source/Main.cpp:51-69 Material, :94-170 Camera, :228-235 Settings, :245-275 Object;
Include/Primitives.h:24-28 Mesh).  All work (glTF parsing, BVH build, flattening) happens in the C++ library.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Sequence, Tuple

import numpy as np

from . import _native as N


class HostError(RuntimeError):
    pass


def _host_check(rc: int, what: str):
    if rc != 0:
        raise HostError(f"{what}: {N.lib().cgpth_last_error().decode()}")


def _f3(v: Sequence[float]):
    return (C.c_float * 3)(float(v[0]), float(v[1]), float(v[2]))


@dataclass
class Material:
    """ref: Source/Main.cpp:51-69"""
    albedo: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    specular: float = 0.0
    refractivity: float = 0.0
    absorption: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    ior: float = 1.0
    emissive: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    intensity: float = 0.0
    is_light: bool = False

    def to_abi(self) -> N.Material:
        m = N.Material()
        m.albedo = _f3(self.albedo); m.specular = self.specular; m.refractivity = self.refractivity
        m.absorption = _f3(self.absorption); m.ior = self.ior; m.emissive = _f3(self.emissive)
        m.intensity = self.intensity; m.is_light = 1 if self.is_light else 0
        return m


@dataclass
class Settings:
    """ref: Source/Main.cpp:228-235 plus render_mode / debug_render_mode (:215-216)"""
    max_ray_depth: int = 5
    next_event_estimation_enabled: bool = True
    cosine_weighted_diffuse_reflection_enabled: bool = True
    russian_roulette_enabled: bool = True
    render_mode: int = N.MODE_ADVANCED
    debug_render_mode: int = N.DEBUG_NONE

    def to_abi(self) -> N.Settings:
        return N.Settings(self.max_ray_depth, int(self.next_event_estimation_enabled),
                          int(self.cosine_weighted_diffuse_reflection_enabled), int(self.russian_roulette_enabled),
                          self.render_mode, self.debug_render_mode)


# the four materials of the shipped scene (ref: Main.cpp:779-782)
REFERENCE_MATERIALS = (
    Material(albedo=(0.2, 0.2, 0.8)),
    Material(albedo=(1.0, 1.0, 1.0)),
    Material(emissive=(1.0, 0.95, 0.8), intensity=10.0, is_light=True),
    Material(albedo=(1.0, 1.0, 1.0), refractivity=1.0, absorption=(0.2, 0.8, 0.8), ior=1.517),
)


class Mesh:
    """ref: Include/Primitives.h:24-28; vertices = n x (pos.xyz, normal.xyz) float32, indices uint32"""

    def __init__(self, handle):
        if not handle:
            raise HostError(N.lib().cgpth_last_error().decode())
        self._h = C.c_void_p(handle)

    @classmethod
    def load_gltf(cls, path: str) -> "Mesh":
        """GLTFLoader::Load (ref: Source/GLTFLoader.cpp:19-89)"""
        return cls(N.lib().cgpth_mesh_load_gltf(path.encode()))

    @classmethod
    def from_arrays(cls, vertices, indices) -> "Mesh":
        v = np.ascontiguousarray(vertices, dtype=np.float32)
        i = np.ascontiguousarray(indices, dtype=np.uint32).ravel()
        assert v.ndim == 2 and v.shape[1] == 6
        return cls(N.lib().cgpth_mesh_from_arrays(v.ctypes.data_as(C.POINTER(N.Vertex)), v.shape[0],
                                                  i.ctypes.data_as(C.POINTER(C.c_uint32)), i.size))

    @classmethod
    def dragon_standin(cls, level: int) -> "Mesh":
        return cls(N.lib().cgpth_mesh_dragon_standin(level))

    @classmethod
    def bumpy_icosphere(cls, level: int, center, radii, bump: float) -> "Mesh":
        return cls(N.lib().cgpth_mesh_bumpy_icosphere(level, _f3(center), _f3(radii), bump))

    def save_gltf(self, path: str):
        _host_check(N.lib().cgpth_mesh_save_gltf(self._h, path.encode()), "save_gltf")

    @property
    def vertices(self) -> np.ndarray:
        L = N.lib()
        n = L.cgpth_mesh_num_vertices(self._h)
        if n == 0:
            return np.zeros((0, 6), np.float32)
        p = C.cast(L.cgpth_mesh_vertices(self._h), C.POINTER(C.c_float))
        return np.ctypeslib.as_array(p, shape=(n, 6)).copy()

    @property
    def indices(self) -> np.ndarray:
        L = N.lib()
        n = L.cgpth_mesh_num_indices(self._h)
        if n == 0:
            return np.zeros(0, np.uint32)
        return np.ctypeslib.as_array(L.cgpth_mesh_indices(self._h), shape=(n,)).copy()

    @property
    def num_triangles(self) -> int:
        return N.lib().cgpth_mesh_num_indices(self._h) // 3

    def close(self):
        if self._h:
            N.lib().cgpth_mesh_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Scene:
    """The parts of the reference's `data` that Render() reads (ref: Main.cpp:209-216,228-235)."""

    def __init__(self, handle=None):
        L = N.lib()
        self._h = C.c_void_p(handle if handle else L.cgpth_scene_new())
        if not self._h:
            raise HostError(L.cgpth_last_error().decode())

    @classmethod
    def reference_layout(cls, mesh: Mesh, mesh_material: int = 3, aspect: float = 16.0 / 9.0,
                         build_option: int = N.BUILD_SAH_INTERVALS) -> "Scene":
        """The shipped scene (ref: Main.cpp:777-819) with `mesh` in place of the dragon."""
        h = N.lib().cgpth_scene_reference_layout(mesh._h, mesh_material, aspect, build_option)
        if not h:
            raise HostError(N.lib().cgpth_last_error().decode())
        return cls(h)

    def add_material(self, m: Material) -> int:
        abi = m.to_abi()
        rc = N.lib().cgpth_scene_add_material(self._h, C.byref(abi))
        if rc < 0:
            raise HostError(N.lib().cgpth_last_error().decode())
        return rc

    def set_material(self, index: int, m: Material):
        abi = m.to_abi()
        _host_check(N.lib().cgpth_scene_set_material(self._h, index, C.byref(abi)), "set_material")

    def add_mesh(self, mesh: Mesh, mat_index: int, build_option: int = N.BUILD_SAH_INTERVALS, device_builder=None) -> int:
        """Object ctor (ref: Main.cpp:247-251).  device_builder: a Renderer whose GPU builds the (bit-identical) tree, any option."""
        if device_builder is not None:
            rc = N.lib().cgpth_scene_add_mesh_device_built_ex(self._h, mesh._h, mat_index, device_builder._ctx, build_option)
        else:
            rc = N.lib().cgpth_scene_add_mesh(self._h, mesh._h, mat_index, build_option)
        if rc < 0:
            raise HostError(N.lib().cgpth_last_error().decode())
        return rc

    def add_sphere(self, center, radius: float, mat_index: int) -> int:
        return N.lib().cgpth_scene_add_sphere(self._h, _f3(center), radius, mat_index)

    def add_plane(self, normal, point, mat_index: int) -> int:
        return N.lib().cgpth_scene_add_plane(self._h, _f3(normal), _f3(point), mat_index)

    def add_light(self, obj_index: int):
        _host_check(N.lib().cgpth_scene_add_light(self._h, obj_index), "add_light")

    def set_camera(self, pos, view_dir, fov_deg: float, aspect: float):
        _host_check(N.lib().cgpth_scene_set_camera(self._h, _f3(pos), _f3(view_dir), fov_deg, aspect), "set_camera")

    def set_settings(self, s: Settings):
        abi = s.to_abi()
        _host_check(N.lib().cgpth_scene_set_settings(self._h, C.byref(abi)), "set_settings")

    def rebuild_bvh(self, obj_index: int, build_option: int, device_builder=None):
        """BVH::Rebuild (ref: BVH.cpp:47-59): re-split over the CURRENT triangle order; device_builder: a Renderer whose GPU does it."""
        if device_builder is not None:
            _host_check(N.lib().cgpth_scene_rebuild_bvh_device(self._h, obj_index, build_option, device_builder._ctx), "rebuild_bvh")
        else:
            _host_check(N.lib().cgpth_scene_rebuild_bvh(self._h, obj_index, build_option), "rebuild_bvh")

    def bvh_info(self, obj_index: int) -> N.BvhInfo:
        info = N.BvhInfo()
        _host_check(N.lib().cgpth_scene_bvh_info(self._h, obj_index, C.byref(info)), "bvh_info")
        return info

    def bvh_export(self, obj_index: int):
        """(nodes[n,8] uint32 words in the reference's 32-byte layout, tri_indices[m] uint32)"""
        info = self.bvh_info(obj_index)
        nodes = np.zeros((info.nodes_used, 8), np.uint32)
        tri = np.zeros(info.num_triangles, np.uint32)
        _host_check(N.lib().cgpth_scene_bvh_export(self._h, obj_index, nodes.ctypes.data_as(C.POINTER(N.BvhNode)),
                                                   tri.ctypes.data_as(C.POINTER(C.c_uint32))), "bvh_export")
        return nodes, tri

    def flatten(self) -> N.SceneDesc:
        desc = N.SceneDesc()
        _host_check(N.lib().cgpth_scene_flatten(self._h, C.byref(desc)), "flatten")
        return desc

    def camera(self) -> N.Camera:
        cam = N.Camera()
        _host_check(N.lib().cgpth_scene_get_camera(self._h, C.byref(cam)), "get_camera")
        return cam

    def settings(self) -> N.Settings:
        s = N.Settings()
        _host_check(N.lib().cgpth_scene_get_settings(self._h, C.byref(s)), "get_settings")
        return s

    def close(self):
        if self._h:
            N.lib().cgpth_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def write_ppm(path: str, pixels: np.ndarray):
    p = np.ascontiguousarray(pixels, dtype=np.uint32)
    _host_check(N.lib().cgpth_write_ppm(path.encode(), p.ctypes.data_as(C.POINTER(C.c_uint32)), p.shape[1], p.shape[0]), "write_ppm")


def write_pfm(path: str, accumulator: np.ndarray, num_accumulated: int):
    a = np.ascontiguousarray(accumulator, dtype=np.float32)
    _host_check(N.lib().cgpth_write_pfm(path.encode(), a.ctypes.data_as(C.POINTER(C.c_float)), num_accumulated,
                                        a.shape[1], a.shape[0]), "write_pfm")


def write_accumulator(path: str, accumulator: np.ndarray, num_accumulated: int):
    a = np.ascontiguousarray(accumulator, dtype=np.float32)
    _host_check(N.lib().cgpth_write_accumulator(path.encode(), a.ctypes.data_as(C.POINTER(C.c_float)), num_accumulated,
                                                a.shape[1], a.shape[0]), "write_accumulator")


def read_accumulator(path: str, width: int, height: int):
    a = np.zeros((height, width, 4), np.float32)
    n = C.c_uint32(0)
    _host_check(N.lib().cgpth_read_accumulator(path.encode(), a.ctypes.data_as(C.POINTER(C.c_float)), C.byref(n), width, height),
                "read_accumulator")
    return a, n.value
