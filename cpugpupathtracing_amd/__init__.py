"""cpugpupathtracing_amd -- MI355X (gfx950) path-tracing hot path behind the reference's Render() boundary.

Layout: csrc/host (C++ scene/BVH/glTF mirror), csrc/device (HIP kernels + C ABI), lib/libcpugpupt.so (built in-tree),
and this thin ctypes layer.  Importing the package does not load the library; the first use does, and fails loudly if the
library is missing.
"""
from ._native import (CTX_FORCE_COLLECTIVE, CTX_GATHER_PEER_COPY, BUILD_NAIVE, BUILD_SAH_INTERVALS, BUILD_SAH_PRIMITIVES, DEBUG_BVH_DEPTH, DEBUG_NONE, DEBUG_RAY_DEPTH,
                      KERNEL_AUTO, KERNEL_MEGAKERNEL, KERNEL_PERSISTENT, KERNEL_WAVEFRONT, MODE_ADVANCED, MODE_BRUTE_FORCE, MODE_COMPARISON,
                      NativeLibraryError)
from .renderer import DeviceError, Renderer
from .scene import REFERENCE_MATERIALS, HostError, Material, Mesh, Scene, Settings

__all__ = [
    "Renderer", "DeviceError", "Scene", "Mesh", "Material", "Settings", "HostError", "NativeLibraryError", "REFERENCE_MATERIALS",
    "BUILD_NAIVE", "BUILD_SAH_INTERVALS", "BUILD_SAH_PRIMITIVES", "MODE_COMPARISON", "MODE_BRUTE_FORCE", "MODE_ADVANCED",
    "DEBUG_NONE", "DEBUG_RAY_DEPTH", "DEBUG_BVH_DEPTH", "KERNEL_AUTO", "KERNEL_MEGAKERNEL", "KERNEL_WAVEFRONT", "KERNEL_PERSISTENT", "CTX_FORCE_COLLECTIVE", "CTX_GATHER_PEER_COPY",
]
