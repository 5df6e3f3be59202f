"""Multi-GPU row tiling of the image and the single framebuffer gather.

One process per GPU.  Rank r of R renders a contiguous band of rows (scene and BVH are replicated by each rank's own
upload, RNG streams are keyed by the GLOBAL pixel index, so the image is identical for any R).  After the render each
rank's float4 accumulator rows go to rank 0 in exactly one collective: torch.distributed.gather, which on the "nccl"
backend is RCCL over xGMI.  No collective happens during tracing.  The same code runs on the "gloo" backend with CPU
tensors (tests/test_distributed.py).

The reference analogue of the consumer side is DX12::CopyToBackBuffer reading data.pixels (ref: Source/DX12.cpp:277).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np


def row_band(height: int, rank: int, world: int) -> Tuple[int, int]:
    """Rows [begin, end) of rank `rank`: bands differ by at most one row; all rows are covered exactly once."""
    assert 0 <= rank < world and height >= world
    base, extra = divmod(height, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def all_bands(height: int, world: int) -> List[Tuple[int, int]]:
    return [row_band(height, r, world) for r in range(world)]


class _DevicePointer:
    """Zero-copy view of a raw device allocation for torch.as_tensor (CUDA array interface v2)."""

    def __init__(self, ptr: int, shape, typestr="<f4"):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (ptr, False), "version": 2}


class FramebufferGather:
    """Gathers the per-rank accumulator bands into rank 0's full (H, W, 4) float32 framebuffer."""

    def __init__(self, width: int, height: int, rank: int, world: int, local_rank: int = 0, device: Optional[str] = None):
        import torch
        self.torch = torch
        self.width, self.height, self.rank, self.world = width, height, rank, world
        self.bands = all_bands(height, world)
        self.max_rows = max(e - b for b, e in self.bands)
        self.device = device if device is not None else f"cuda:{local_rank}"
        # every rank sends max_rows rows (shorter bands are zero padded) so one equal-count gather suffices
        self.send = torch.zeros((self.max_rows, width, 4), dtype=torch.float32, device=self.device)
        self.recv = ([torch.zeros_like(self.send) for _ in range(world)] if rank == 0 else None)
        self.full = torch.zeros((height, width, 4), dtype=torch.float32, device=self.device) if rank == 0 else None

    def gather_tensor(self, band):
        """band: this rank's (rows, W, 4) float32 tensor on self.device.  Returns the full framebuffer on rank 0."""
        import torch.distributed as dist
        b, e = self.bands[self.rank]
        assert tuple(band.shape) == (e - b, self.width, 4), (tuple(band.shape), (e - b, self.width, 4))
        self.send[: e - b].copy_(band)
        dist.gather(self.send, gather_list=self.recv, dst=0)       # the ONE collective: float4 rows -> rank 0
        if self.rank != 0:
            return None
        for r, (rb, re) in enumerate(self.bands):
            self.full[rb:re].copy_(self.recv[r][: re - rb])
        return self.full

    def gather(self, renderer):
        """Gathers straight from the renderer's device accumulator (no host round trip)."""
        ptr, nbytes = renderer.accumulator_device_ptr()
        b, e = self.bands[self.rank]
        assert nbytes == (e - b) * self.width * 16
        band = self.torch.as_tensor(_DevicePointer(ptr, (e - b, self.width, 4)), device=self.device)
        return self.gather_tensor(band)


def pack_pixels(accumulator: np.ndarray, num_accumulated: int) -> np.ndarray:
    """Vec4ToUint(accumulator / n) (ref: Include/MathLib.h:144-152, Main.cpp:741) for the gathered framebuffer on the
    host; matches the device packing bit for bit (truncation, no gamma, negative clamped to 0)."""
    a = accumulator[..., :3].astype(np.float32) / np.float32(num_accumulated)
    v = np.float32(255.0) * np.minimum(np.float32(1.0), a)
    v = np.where(v < 0, np.float32(0.0), v).astype(np.int32).astype(np.uint32) & 0xFF
    return (np.uint32(255) << 24) + (v[..., 2] << 16) + (v[..., 1] << 8) + v[..., 0]
