#!/usr/bin/env python3
"""bench.py -- Mrays/s and ms/frame of the path-tracing hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A *step* is one pass of the hot path over the whole workload: a full `spp`-sample render of the frame (cgpt_render with
n_samples = spp, i.e. `spp` reference Render() calls, ref: Source/Main.cpp:691-755).  A *ray* is one IntersectScene call
(ref: Main.cpp:301).  Workload at N = 1 (config.workload): the configuration BASELINE.json's metric is quoted on --
glass dragon stand-in (81 920 triangles, SAH-intervals BVH, loaded through the glTF path), 1920x1080, 256 spp,
TracePathAdvanced with the reference's default settings.  Inputs (scene, BVH) are resident in HBM before the timed region.

For N > 1 the image is row-tiled over the ranks in interleaved 4-row bands (one process per GPU, scene replicated) and the
float4 accumulator rows are gathered to rank 0 with ONE RCCL collective per step (torch.distributed gather on the nccl backend = RCCL over xGMI);
the gather is inside the timed region.  Total work is fixed as N grows ("scaling": "strong").

Extra objects on the JSON line: `roofline` (algorithmic bytes of the traversal per launch / measured kernel time vs the
8 TB/s HBM peak; formula in DESIGN.md) and `cpu_baseline` (the CPU oracle, the port of the reference's ThreadPool path,
timed on this box's host cores on a bounded sample; reported, not a target).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--level", type=int, default=6, help="icosphere level of the dragon stand-in (6 = 81 920 triangles)")
    ap.add_argument("--material", type=int, default=3, help="material of the mesh (3 = the reference's glass, Main.cpp:782)")
    ap.add_argument("--kernel", choices=["auto", "megakernel", "wavefront"], default="auto")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="target CPU time of the cpu_baseline sample (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=16, help="threads of the cpu_baseline leg (the box's CPU share per GPU)")
    ap.add_argument("--band-rows", type=int, default=4, help="rows per interleaved band for N > 1 (measured on 8-way shares of the 1080p frame: 8 rows 14.0-15.8 ms per rank, 4 rows 14.7-15.5, 1 row 15.3-15.4)")
    ap.add_argument("--rehearse-gloo", action="store_true", help="N > 1 rehearsal on a 1-GPU box: every rank renders on cuda:0 and the "
                    "gather runs on the gloo backend through host tensors (exercises tiling, gather, reorder and timing; not a measurement)")
    ap.add_argument("--force-collective", action="store_true", help="N = 1 only: initialise RCCL and run the framebuffer gather anyway "
                    "(world size 1), to exercise the collective path of N > 1 on a one-GPU box")
    ap.add_argument("--simulate-rank", type=int, default=None, help="rehearsal on one GPU: render only rank R's bands of a --simulate-world job (no collective)")
    ap.add_argument("--simulate-world", type=int, default=8)
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x12345678)
    return ap.parse_args()


def algorithmic_bytes(st, width, rows, spp):
    """sum over rays and mesh objects of [32 (root) + 64*N_inner + 40*N_tri] + 12*N_closest_hits + 32*W*H*P.
    The reference scene has two mesh objects (dragon stand-in, ground), so the root term is 2 * 32 per ray; P = 1
    accumulate pass per launch here (the sample loop is inside the kernel, so the accumulator is read and written once)."""
    n_mesh_objects = 2
    return (32 * n_mesh_objects * st.traced_rays + 64 * st.inner_steps + 40 * st.tri_tests + 12 * st.closest_hits
            + 32 * width * rows * 1)


def cpu_baseline(args, vertices, indices, aspect):
    """The oracle (CPU port of the reference's ThreadPool path) on this box's host cores, bounded sample."""
    import oracle as O
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = max(1, min(cores, args.cpu_threads))     # the GPU box's CPU share for one GPU is 16 cores
    o = O.OracleScene()
    import cpugpupathtracing_amd as P
    for m in P.REFERENCE_MATERIALS:
        o.add_material(m.albedo, m.specular, m.refractivity, m.absorption, m.ior, m.emissive, m.intensity, m.is_light)
    ground_v = np.array([[-1000, -3, 1000, 0, 1, 0], [-1000, -3, -1000, 0, 1, 0], [1000, -3, -1000, 0, 1, 0], [1000, -3, 1000, 0, 1, 0]], np.float32)
    o.add_mesh(vertices, indices, args.material, O.BUILD_SAH_INTERVALS)
    o.add_mesh(ground_v, np.array([0, 1, 2, 2, 3, 0], np.uint32), 1, O.BUILD_SAH_INTERVALS)
    for c in ((10.0, 10.0, 10.0), (-10.0, 10.0, -10.0)):
        o.add_light(o.add_sphere(c, 5.0, 2))
    o.set_camera((0, 0, 8), (0, 0, -1), 60.0, aspect)
    # calibrate with one frame, then run enough frames to fill the budget
    t0 = time.perf_counter()
    o.render(args.width, args.height, 1, O.MODE_ADVANCED, O.DEBUG_NONE, O.RNG_PIXEL_PCG, args.seed, nthreads=cores)
    o.render(args.width, args.height, 1, O.MODE_ADVANCED, O.DEBUG_NONE, O.RNG_PIXEL_PCG, args.seed, nthreads=cores)
    t1 = (time.perf_counter() - t0) / 2
    frames = max(1, min(256, int(args.cpu_seconds / max(t1, 1e-3))))
    o.reset_stats()
    t0 = time.perf_counter()
    o.render(args.width, args.height, frames, O.MODE_ADVANCED, O.DEBUG_NONE, O.RNG_PIXEL_PCG, args.seed, nthreads=cores)
    dt = time.perf_counter() - t0
    rays = o.stats().traced_rays
    return {"value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"{frames} frame(s) (spp) of the same {args.width}x{args.height} scene, same seeds, {dt:.1f} s; "
                      f"ms_per_frame={1e3 * dt / frames:.1f}"}


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                         "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")

    import torch
    import cpugpupathtracing_amd as P
    from cpugpupathtracing_amd import distributed as D

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the render path")
    if args.rehearse_gloo:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    collective = world > 1 or args.force_collective
    if collective:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if args.rehearse_gloo:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    # ---- scene: synthetic dragon stand-in written as glTF and loaded back through the glTF path ----
    aspect = args.width / args.height
    with tempfile.TemporaryDirectory() as tmp:
        gen = P.Mesh.dragon_standin(args.level)
        path = os.path.join(tmp, f"standin_l{args.level}_r{rank}.gltf")
        gen.save_gltf(path)
        mesh = P.Mesh.load_gltf(path)
    scene = P.Scene.reference_layout(mesh, args.material, aspect, P.BUILD_SAH_INTERVALS)
    n_tris = mesh.num_triangles

    renderer = P.Renderer(local_rank)
    renderer.upload(scene)
    kernel = {"auto": P.KERNEL_AUTO, "megakernel": P.KERNEL_MEGAKERNEL, "wavefront": P.KERNEL_WAVEFRONT}[args.kernel]
    # N > 1: 8-row bands dealt round-robin over the ranks, so every GPU gets the same mix of sky, mesh and ground rows
    interleave = (args.band_rows, world, rank) if world > 1 else None
    n_rows = len(D.interleaved_rows(args.height, rank, world, args.band_rows)) if world > 1 else args.height
    if args.simulate_rank is not None and world == 1:
        interleave = (args.band_rows, args.simulate_world, args.simulate_rank)
        n_rows = len(D.interleaved_rows(args.height, args.simulate_rank, args.simulate_world, args.band_rows))
    gather = None
    if collective:
        gather = D.FramebufferGather(args.width, args.height, rank, world, local_rank, band_rows=args.band_rows,
                                     device="cpu" if args.rehearse_gloo else None)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step(counters=False):
        renderer.reset_accumulator()
        renderer.render(args.width, args.height, args.spp, seed=args.seed, interleave=interleave, kernel=kernel, counters=counters)
        if gather is not None:
            if args.rehearse_gloo:
                step.full = gather.gather_tensor(torch.from_numpy(renderer.accumulator()))
            else:
                step.full = gather.gather(renderer)

    # ---- warmup (untimed); the first warmup step also collects the traversal counters for the roofline ----
    renderer.reset_stats()
    step(counters=True)
    st_count = renderer.stats()
    b_alg = algorithmic_bytes(st_count, args.width, n_rows, args.spp)
    rays_per_step_local = st_count.traced_rays
    for _ in range(max(0, args.warmup - 1)):
        step()

    # ---- timed region: exactly K steps ----
    renderer.reset_stats()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    st = renderer.stats()

    red_dev = "cpu" if args.rehearse_gloo else "cuda"
    t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    rays = torch.tensor([float(st.traced_rays)], dtype=torch.float64, device=red_dev)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    total_rays = float(rays.item())

    if (args.rehearse_gloo and world > 1 or args.force_collective and world == 1) and rank == 0:
        # rehearsal check: the gathered, re-ordered framebuffer equals a single-context render of the whole frame
        renderer.reset_accumulator()
        renderer.render(args.width, args.height, args.spp, seed=args.seed, kernel=kernel)
        same = np.array_equal(renderer.accumulator().view(np.uint32), step.full.cpu().numpy().view(np.uint32))
        print(f"[rehearsal] gathered framebuffer identical to the single-GPU render: {same}", file=sys.stderr, flush=True)
        if not same:
            raise SystemExit("rehearsal mismatch")

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        # roofline of the dominant kernel (the megakernel, or the wavefront pipeline's trace kernel): algorithmic bytes per
        # launch = bytes/ray (SURVEY 8d formula over this rank's counters) x rays per launch; duration = that kernel's own
        # hipEvent time on the stream it was launched on, averaged over its launches in the timed region
        dominant = "wf_trace" if (args.kernel == "wavefront" or (args.kernel == "auto" and st.dominant_launches > args.steps)) else "megakernel"
        launches_per_step = st.dominant_launches / max(1, args.steps)
        bytes_per_ray = b_alg / max(1, rays_per_step_local)
        rays_per_launch = (st.traced_rays / max(1, args.steps)) / max(1.0, launches_per_step)
        ms_per_launch = st.dominant_ms / max(1, st.dominant_launches)
        achieved_gbs = bytes_per_ray * rays_per_launch / (ms_per_launch * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(REPO, "profiles", "hbm_traffic.json")
        key = f"{args.width}x{args.height}x{args.spp}_l{args.level}_m{args.material}_{dominant}_n{world}"
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(key)
            except Exception:
                traffic = None
        out = {
            "metric": f"Mrays/sec @ {args.width}x{args.height}, {args.spp} spp (dragon glTF stand-in)",
            "value": round(total_rays / elapsed / 1e6, 3),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "ms_per_frame": round(ms_per_step / args.spp, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"glass dragon stand-in (bumpy icosphere level {args.level}, {n_tris} tris, SAH-intervals BVH, via glTF) in the "
                            f"reference scene layout (Main.cpp:777-819), material {args.material}, {args.width}x{args.height}, {args.spp} spp, "
                            "TracePathAdvanced defaults (NEE, RR, cosine, max depth 5)",
                "kernel": args.kernel, "rays_per_step": int(total_rays / args.steps), "rows_per_gpu": n_rows,
                "parallelism": (f"{args.band_rows}-row bands interleaved over {world} GPUs + 1 RCCL gather/step" if world > 1 else "1 GPU"),
            },
            "roofline": {
                "bound": "hbm", "achieved": round(achieved_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved_gbs / HBM_PEAK_GBS, 5), "traffic": traffic,
                "kernel": dominant, "algorithmic_bytes_per_ray": round(bytes_per_ray, 2),
                "rays_per_launch": int(rays_per_launch), "algorithmic_bytes_per_launch": int(bytes_per_ray * rays_per_launch),
                "kernel_ms_per_launch": round(ms_per_launch, 4), "launches_per_step": round(launches_per_step, 1),
                "render_ms_per_step": round(st.kernel_ms / max(1, args.steps), 3),
                # the same algorithmic bytes over the whole render (all kernels of all overlapping batches, context-stream
                # hipEvents): the machine-level rate; `achieved` above charges every trace launch its full wall duration
                # although up to 8 launches share the chip
                "pipeline_achieved": round(b_alg / (st.kernel_ms / max(1, args.steps) * 1e-3) / 1e9, 2),
                "pipeline_frac": round(b_alg / (st.kernel_ms / max(1, args.steps) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
            },
        }
        if args.cpu_seconds > 0 and world == 1:
            sys.path.insert(0, os.path.join(REPO, "oracle"))
            out["cpu_baseline"] = cpu_baseline(args, mesh.vertices, mesh.indices, aspect)
        print(json.dumps(out), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
