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


def interleaved_rows(height: int, rank: int, world: int, band_rows: int = 8) -> np.ndarray:
    """Global rows of rank `rank` under interleaved tiling: bands of `band_rows` rows dealt round-robin over the ranks
    (band b goes to rank b % world).  Same enumeration as cgpt_render_params.interleave_* (include/cpugpupt_abi.h)."""
    assert 0 <= rank < world and band_rows > 0
    rows = []
    first = rank * band_rows
    while first < height:
        rows.extend(range(first, min(first + band_rows, height)))
        first += world * band_rows
    return np.asarray(rows, dtype=np.int64)


def all_bands(height: int, world: int) -> List[Tuple[int, int]]:
    return [row_band(height, r, world) for r in range(world)]


class _DevicePointer:
    """Zero-copy view of a raw device allocation for torch.as_tensor (CUDA array interface v2)."""

    def __init__(self, ptr: int, shape, typestr="<f4"):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (ptr, False), "version": 2}


class FramebufferGather:
    """Gathers the per-rank accumulator bands into rank 0's full (H, W, 4) float32 framebuffer.

    band_rows = None: contiguous bands (row_band).  band_rows = h: interleaved bands of h rows (interleaved_rows), which
    spreads the expensive rows (the mesh) and the cheap ones (sky) evenly over the GPUs."""

    def __init__(self, width: int, height: int, rank: int, world: int, local_rank: int = 0, device: Optional[str] = None,
                 band_rows: Optional[int] = None):
        import torch
        self.torch = torch
        self.width, self.height, self.rank, self.world = width, height, rank, world
        self.band_rows = band_rows
        self.device = device if device is not None else f"cuda:{local_rank}"
        if band_rows is None:
            self.bands = all_bands(height, world)
            self.row_index = [torch.arange(b, e, device=self.device) for b, e in self.bands]
        else:
            self.row_index = [torch.as_tensor(interleaved_rows(height, r, world, band_rows), device=self.device) for r in range(world)]
        self.n_rows = [int(ix.numel()) for ix in self.row_index]
        self.max_rows = max(self.n_rows)
        # every rank sends max_rows rows (shorter bands are zero padded) so one equal-count gather suffices
        self.send = torch.zeros((self.max_rows, width, 4), dtype=torch.float32, device=self.device)
        self.recv = ([torch.zeros_like(self.send) for _ in range(world)] if rank == 0 else None)
        self.full = torch.zeros((height, width, 4), dtype=torch.float32, device=self.device) if rank == 0 else None
        self.copied = torch.cuda.Event() if str(self.device).startswith("cuda") else None   # marks the end of the copy out of the accumulator

    def gather_tensor(self, band):
        """band: this rank's (rows, W, 4) float32 tensor on self.device.  Returns the full framebuffer on rank 0."""
        import torch.distributed as dist
        n = self.n_rows[self.rank]
        assert tuple(band.shape) == (n, self.width, 4), (tuple(band.shape), (n, self.width, 4))
        self.send[:n].copy_(band)
        if self.copied is not None:
            self.copied.record()
        dist.gather(self.send, gather_list=self.recv, dst=0)       # the ONE collective: float4 rows -> rank 0
        if self.rank != 0:
            return None
        for r, ix in enumerate(self.row_index):                      # local reorder on rank 0
            self.full.index_copy_(0, ix, self.recv[r][: self.n_rows[r]])
        return self.full

    def gather(self, renderer):
        """Gathers straight from the renderer's device accumulator (no host round trip)."""
        ptr, nbytes = renderer.accumulator_device_ptr()
        n = self.n_rows[self.rank]
        assert nbytes == n * self.width * 16
        band = self.torch.as_tensor(_DevicePointer(ptr, (n, self.width, 4)), device=self.device)
        full = self.gather_tensor(band)
        # the accumulator belongs to the renderer's own HIP stream, which knows nothing of torch's: the copy out of it (first thing
        # gather_tensor queues) must have finished before the caller resets or renders into it again
        self.copied.synchronize()
        return full


def pack_pixels(accumulator: np.ndarray, num_accumulated: int) -> np.ndarray:
    """Vec4ToUint(accumulator / n) (ref: Include/MathLib.h:144-152, Main.cpp:741) for the gathered framebuffer on the
    host; matches the device packing bit for bit (truncation, no gamma, negative clamped to 0)."""
    a = accumulator[..., :3].astype(np.float32) / np.float32(num_accumulated)
    v = np.float32(255.0) * np.minimum(np.float32(1.0), a)
    v = np.where(v < 0, np.float32(0.0), v).astype(np.int32).astype(np.uint32) & 0xFF
    return (np.uint32(255) << 24) + (v[..., 2] << 16) + (v[..., 1] << 8) + v[..., 0]
