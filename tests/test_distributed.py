"""The N > 1 path on CPU: world_size-2 (and 3) gloo runs of the row tiling + single framebuffer gather.

The band renderer here is the oracle (tests may use it); what is under test is the product's sharding plumbing
(cpugpupathtracing_amd.distributed): bands cover the image, RNG keyed by global pixel index makes the tiled image
bit-identical to the single-rank image, one gather reassembles it on rank 0.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, W, H, spp, out_path, band_rows=None):
    sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle")); sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle as O
    from cpugpupathtracing_amd import distributed as D
    from scenes import reference_layout_pair, standin_mesh
    v, i = standin_mesh(2)
    o, _ = reference_layout_pair(v, i, 3, aspect=W / H)
    if band_rows is None:
        rows = D.row_band(H, rank, world)
        o.render(W, H, spp, O.MODE_ADVANCED, O.DEBUG_NONE, O.RNG_PIXEL_PCG, 0x12345678, nthreads=1, rows=rows)
        band = torch.from_numpy(o.accumulator()[rows[0]:rows[1]].copy())
    else:   # interleaved bands: the oracle renders whole frames, the rank keeps only its own rows
        o.render(W, H, spp, O.MODE_ADVANCED, O.DEBUG_NONE, O.RNG_PIXEL_PCG, 0x12345678, nthreads=1)
        band = torch.from_numpy(o.accumulator()[D.interleaved_rows(H, rank, world, band_rows)].copy())
    g = D.FramebufferGather(W, H, rank, world, device="cpu", band_rows=band_rows)
    full = g.gather_tensor(band)
    if rank == 0:
        np.save(out_path, full.numpy())
    else:
        assert full is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,H,band_rows", [(2, 48, None), (3, 50, None), (2, 44, 8), (3, 50, 8)])
def test_row_tiled_gather_equals_single_rank_image(tmp_path, world, H, band_rows):
    import oracle as O
    from scenes import reference_layout_pair, standin_mesh
    W, spp = 64, 2
    out = str(tmp_path / "full.npy")
    mp.spawn(_worker, args=(world, _free_port(), W, H, spp, out, band_rows), nprocs=world, join=True)
    got = np.load(out)
    v, i = standin_mesh(2)
    o, _ = reference_layout_pair(v, i, 3, aspect=W / H)
    o.render(W, H, spp, O.MODE_ADVANCED, O.DEBUG_NONE, O.RNG_PIXEL_PCG, 0x12345678, nthreads=2)
    want = o.accumulator()
    assert got.shape == want.shape
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
