"""Renderer: the reference's Render() / ResetAccumulator() / data.accumulator / data.pixels / data.stats surface
(ref: Source/Main.cpp:200-243,691-755) on top of the HIP library's C ABI (include/cpugpupt_abi.h).

No CPU path: constructing a Renderer without a gfx950 device raises DeviceError.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _native as N
from .scene import Scene, Settings


class DeviceError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"[cgpt status {code}] {message}")
        self.code = code


class Renderer:
    def __init__(self, device=0, flags: int = 0):
        """device: one HIP device id, or a sequence of up to 8 ids for ONE context that spreads every frame over those GPUs
        (interleaved row bands, one RCCL exchange of the float4 bands per read-back; cgpt_ctx_create).  flags: N.CTX_*."""
        self.L = N.lib()
        self._ctx = C.c_void_p()
        devices = [int(device)] if isinstance(device, (int, np.integer)) else [int(d) for d in device]
        ids = (C.c_int * len(devices))(*devices)
        rc = self.L.cgpt_ctx_create(ids, len(devices), flags, C.byref(self._ctx))
        if rc != 0:
            raise DeviceError(rc, self.L.cgpt_last_error(None).decode())
        self.devices = devices
        self.is_group = len(devices) > 1 or bool(flags & N.CTX_FORCE_COLLECTIVE)
        self.scene: Optional[Scene] = None
        self.width = self.height = 0
        self.rows: Tuple[int, int] = (0, 0)
        self.interleave: Optional[Tuple[int, int, int]] = None
        self.n_rows = 0               # rows of this context's band
        self.num_accumulated = 0      # ref: Main.cpp:205

    def _check(self, rc: int):
        if rc != 0:
            raise DeviceError(rc, self.L.cgpt_last_error(self._ctx).decode())

    def set_stream(self, hip_stream: int):
        self._check(self.L.cgpt_set_stream(self._ctx, C.c_void_p(hip_stream)))

    def upload(self, scene: Scene):
        desc = scene.flatten()
        self._check(self.L.cgpt_scene_upload(self._ctx, C.byref(desc)))
        self.scene = scene

    def update_materials(self, scene: Scene):
        desc = scene.flatten()
        self._check(self.L.cgpt_scene_update_materials(self._ctx, desc.materials, desc.n_materials))

    def render(self, width: int, height: int, n_samples: int = 1, seed: int = 0x12345678, rows: Optional[Tuple[int, int]] = None,
               kernel: int = N.KERNEL_AUTO, counters: bool = False, settings: Optional[Settings] = None,
               interleave: Optional[Tuple[int, int, int]] = None):
        """Render() x n_samples (ref: Main.cpp:691-755).  Accumulates; call reset_accumulator() to start over.
        rows = (begin, end): a contiguous band.  interleave = (band_rows, count, index): every count-th band of band_rows
        rows starting at band `index` (load-balanced multi-GPU tiling); the band is stored compactly in that order."""
        assert self.scene is not None, "upload a scene first"
        assert rows is None or interleave is None
        r0, r1 = rows if rows is not None else (0, height)
        il = interleave if interleave is not None else (0, 0, 0)
        if (width, height, (r0, r1), interleave) != (self.width, self.height, self.rows, self.interleave):
            self.num_accumulated = 0          # the library re-allocates (zeroed) on a size/band change
        cam = self.scene.camera()
        st = settings.to_abi() if settings is not None else self.scene.settings()
        p = N.RenderParams(width, height, r0, r1, self.num_accumulated, n_samples, seed & 0xFFFFFFFF, kernel,
                           N.RENDER_COUNTERS if counters else 0, il[0], il[1], il[2])
        self._check(self.L.cgpt_render(self._ctx, C.byref(cam), C.byref(st), C.byref(p)))
        self.width, self.height, self.rows, self.interleave = width, height, (r0, r1), interleave
        if interleave is None:
            self.n_rows = r1 - r0
        else:
            from .distributed import interleaved_rows
            self.n_rows = len(interleaved_rows(height, interleave[2], interleave[1], interleave[0]))
        self.num_accumulated += n_samples

    def reset_accumulator(self):
        self._check(self.L.cgpt_reset_accumulator(self._ctx))
        self.num_accumulated = 0

    def accumulator(self) -> np.ndarray:
        """float4 running sums of this context's row band: (rows, width, 4)"""
        out = np.empty((self.n_rows, self.width, 4), np.float32)
        self._check(self.L.cgpt_read_accumulator(self._ctx, out.ctypes.data_as(C.POINTER(C.c_float)), out.size))
        return out

    def load_accumulator(self, acc: np.ndarray, num_accumulated: int, width: int, height: int,
                         rows: Optional[Tuple[int, int]] = None, interleave: Optional[Tuple[int, int, int]] = None):
        """Restores a saved accumulator band and its sample count (checkpoint / resume: data.accumulator + data.num_accumulated,
        ref: Main.cpp:204-205); the next render() continues at sample `num_accumulated`, bit-identical to an uninterrupted run."""
        a = np.ascontiguousarray(acc, np.float32)
        r0, r1 = rows if rows is not None else (0, height)
        il = interleave if interleave is not None else (0, 0, 0)
        p = N.RenderParams(width, height, r0, r1, 0, 0, 0, 0, 0, il[0], il[1], il[2])
        self._check(self.L.cgpt_write_accumulator(self._ctx, C.byref(p), a.ctypes.data_as(C.POINTER(C.c_float)), a.size, num_accumulated))
        self.width, self.height, self.rows, self.interleave = width, height, (r0, r1), interleave
        if interleave is None:
            self.n_rows = r1 - r0
        else:
            from .distributed import interleaved_rows
            self.n_rows = len(interleaved_rows(height, interleave[2], interleave[1], interleave[0]))
        self.num_accumulated = num_accumulated

    def set_tuning(self, **knobs: int):
        """cgpt_set_tuning: wavefront knobs for this context (pools=1, batch=16, ...); never changes results."""
        for name, value in knobs.items():
            self._check(self.L.cgpt_set_tuning(self._ctx, name.encode(), int(value)))

    def measure_issue_rate(self, kind: int = 0, waves_per_simd: int = 6, iters: int = 20000) -> Tuple[float, float]:
        """Measured vector-instruction issue rate of the device (wave64 instructions / s over the chip) and the launch's ms."""
        rate = C.c_double(); ms = C.c_double()
        self._check(self.L.cgpt_measure_issue_rate(self._ctx, kind, waves_per_simd, iters, C.byref(rate), C.byref(ms)))
        return rate.value, ms.value

    def pixels(self) -> np.ndarray:
        out = np.empty((self.n_rows, self.width), np.uint32)
        self._check(self.L.cgpt_read_pixels(self._ctx, out.ctypes.data_as(C.POINTER(C.c_uint32)), out.size))
        return out

    def accumulator_device_ptr(self) -> Tuple[int, int]:
        ptr = C.c_void_p(); nbytes = C.c_size_t()
        self._check(self.L.cgpt_accumulator_device_ptr(self._ctx, C.byref(ptr), C.byref(nbytes)))
        return ptr.value, nbytes.value

    def stats(self) -> N.Stats:
        s = N.Stats()
        self._check(self.L.cgpt_get_stats(self._ctx, C.byref(s)))
        return s

    def reset_stats(self):
        self._check(self.L.cgpt_reset_stats(self._ctx))

    def intersect_rays(self, origins, dirs, tmax=None):
        """IntersectScene on a ray batch (ref: Main.cpp:299-316): returns t, obj_idx, tri_idx, bvh_depth"""
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        n = o.shape[0]
        tm = None if tmax is None else np.ascontiguousarray(tmax, np.float32)
        t = np.empty(n, np.float32); obj = np.empty(n, np.uint32); tri = np.empty(n, np.uint32); dep = np.empty(n, np.uint32)
        fp, up = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
        self._check(self.L.cgpt_intersect_rays(self._ctx, o.ctypes.data_as(fp), d.ctypes.data_as(fp),
                                               tm.ctypes.data_as(fp) if tm is not None else None, n, t.ctypes.data_as(fp),
                                               obj.ctypes.data_as(up), tri.ctypes.data_as(up), dep.ctypes.data_as(up)))
        return t, obj, tri, dep

    def build_bvh(self, triangles, n_tris: int, build_option: int = N.BUILD_SAH_INTERVALS, initial_tri_indices=None):
        """BVH::Build (or, with initial_tri_indices = the current m_tri_indices, BVH::Rebuild) on the GPU for any BuildOption
        (ref: Source/BVH.cpp:11-59,204-297): returns (nodes[n,8] uint32 words in the reference's 32-byte layout, tri_indices[n_tris],
        max_depth, total_area).  triangles: ctypes pointer to n_tris cgpt_triangle (e.g. SceneDesc.triangles + tri_offset)."""
        nodes = np.zeros((max(2 * n_tris - 1, 1), 8), np.uint32)
        tri = np.zeros(n_tris, np.uint32)
        n_nodes = C.c_uint32(); depth = C.c_uint32(); area = C.c_float()
        init = None
        if initial_tri_indices is not None:
            init_arr = np.ascontiguousarray(initial_tri_indices, np.uint32)
            assert init_arr.shape == (n_tris,)
            init = init_arr.ctypes.data_as(C.POINTER(C.c_uint32))
        self._check(self.L.cgpt_bvh_build_ex(self._ctx, triangles, n_tris, build_option, init, nodes.ctypes.data_as(C.POINTER(N.BvhNode)), C.byref(n_nodes),
                                             tri.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(depth), C.byref(area)))
        return nodes[:n_nodes.value].copy(), tri, depth.value, area.value

    def synchronize(self):
        self._check(self.L.cgpt_synchronize(self._ctx))

    def close(self):
        if self._ctx:
            self.L.cgpt_ctx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
