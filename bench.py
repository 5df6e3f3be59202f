#!/usr/bin/env python3
"""bench.py -- Mrays/s and ms/frame of the path-tracing hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A *step* is one pass of the hot path over the whole workload: a full `spp`-sample render of the frame (cgpt_render with
n_samples = spp, i.e. `spp` reference Render() calls, ref: Source/Main.cpp:691-755).  A *ray* is one IntersectScene call
(ref: Main.cpp:301).  Workload at N = 1 (config.workload): the configuration BASELINE.json's metric is quoted on --
glass dragon stand-in (81 920 triangles, SAH-intervals BVH, loaded through the glTF path), 1920x1080, 256 spp,
TracePathAdvanced with the reference's default settings.  Inputs (scene, BVH) are resident in HBM before the timed region.

For N > 1 the image is row-tiled over the GPUs in interleaved 4-row bands (scene replicated) and the float4 accumulator rows
are gathered to GPU 0 with ONE RCCL exchange per step, inside the timed region.  Total work is fixed as N grows ("scaling":
"strong").  Two hosts for the same tiling:
  * launched as a plain command (`python bench.py --gpus N`, WORLD_SIZE unset): ONE process, the C ABI's multi-device context
    (cgpt_ctx_create over N devices, csrc/device/multi_gpu.hip: a host worker per device, ncclCommInitAll, one grouped
    ncclSend / ncclRecv exchange + a row-reorder kernel) -- what a C++ host following INTEGRATION.md gets;
  * launched under torch.distributed.run (WORLD_SIZE set): one process per GPU, torch.distributed gather on the nccl backend.
`--share-gpu` (in-process only) puts all N ranks on GPU 0 with peer copies in place of RCCL, and `--force-collective` sends N = 1
through the in-process RCCL path: both are rehearsals of the N > 1 plumbing on a one-GPU box and check the gathered image.

Extra objects on the JSON line:
`roofline` -- of the dominant kernel (wf_trace).  The BVH working set is cache resident, so the kernel's roof is vector-
instruction ISSUE, not HBM: `peak` is MEASURED in this process by cgpt_measure_issue_rate (independent
v_mul_f32 streams at the kernel's own waves/SIMD on every CU), `achieved` = the kernel's VALU wave-instructions per step (rocprofv3
SQ_INSTS_VALU, profiles/pmc_counts.json, same command line) / the summed duration of its launches, timed with hipEvents in a
single-pool pass (batches one after the other, so a launch has the chip to itself; with the production setting several batches
overlap and a launch's wall time is shared).  HBM appears as `hbm_frac` = counter bytes (`traffic`) / kernel time / 8 TB/s; the
SURVEY-8d algorithmic bytes are reported as a rate (`algorithmic_gbs`), never as a fraction of HBM peak: most are served by L2 / LDS.
`cpu_baseline` -- the CPU oracle (the port of the reference's ThreadPool path) timed on this box's host cores on a bounded
sample; reported, not a target.
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
    ap.add_argument("--kernel", choices=["auto", "policy", "megakernel", "wavefront", "persistent"], default="auto",
                    help="auto: CGPT_KERNEL_AUTO, or the kernel a --config names; policy: CGPT_KERNEL_AUTO even under --config")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="target CPU time of the cpu_baseline sample (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=16, help="threads of the cpu_baseline leg (the box's CPU share per GPU)")
    ap.add_argument("--band-rows", type=int, default=4, help="rows per interleaved band for N > 1 (measured on 8-way shares of the 1080p frame, ms per rank, profiles/r02/band_rows_rank_shares.txt: 2 rows 13.7-14.3, 4 rows 13.7-14.2, 8 rows 13.2-14.5, 16 rows 12.8-15.2)")
    ap.add_argument("--rehearse-gloo", action="store_true", help="N > 1 rehearsal on a 1-GPU box: every rank renders on cuda:0 and the "
                    "gather runs on the gloo backend through host tensors (exercises tiling, gather, reorder and timing; not a measurement)")
    ap.add_argument("--force-collective", action="store_true", help="N = 1 only: run the framebuffer gather anyway, to exercise the collective path of "
                    "N > 1 on a one-GPU box (in-process: the multi-device context's RCCL exchange with one rank; under torchrun: torch.distributed nccl)")
    ap.add_argument("--share-gpu", action="store_true", help="in-process N > 1 rehearsal on a one-GPU box: all N ranks of the multi-device context on "
                    "GPU 0, peer copies in place of RCCL (CGPT_CTX_GATHER_PEER_COPY); checks the gathered image; not a measurement")
    ap.add_argument("--simulate-rank", type=int, default=None, help="rehearsal on one GPU: render only rank R's bands of a --simulate-world job (no collective)")
    ap.add_argument("--simulate-world", type=int, default=8)
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x12345678)
    ap.add_argument("--config", choices=["C1", "C2", "C3", "C4", "C5"], default=None,
                    help="a BASELINE.json configuration: C1 256x256 1 spp diffuse | C2 720p 64 spp diffuse+specular megakernel | "
                         "C3 1080p 256 spp glass wavefront (the default workload) | C4 1.3 M triangles 1080p 1024 spp, one rank's share of 8 | "
                         "C5 the same scene 4K 4096 spp, one rank's share of 8 (pick the rank with --simulate-rank)")
    ap.add_argument("--mode", choices=["advanced", "brute", "comparison"], default="advanced",
                    help="render_mode (ref: Main.cpp:172-178): TracePathAdvanced | TracePath (brute force) | the reference's default split screen")
    ap.add_argument("--pools", type=int, default=0, help="sample batches in flight in the timed region (0 = the library default)")
    ap.add_argument("--no-roofline-pass", action="store_true", help="skip the single-pool pass and the issue-rate measurement")
    ap.add_argument("--issue-table", action="store_true", help="print the measured issue rates for every kind and 1..8 waves/SIMD to stderr")
    a = ap.parse_args()
    a.scene = "standin"
    if a.config == "C1":
        a.width, a.height, a.spp, a.level, a.material = 256, 256, 1, 6, 1
    elif a.config == "C2":
        a.width, a.height, a.spp, a.level, a.material = 1280, 720, 64, 6, 4
        a.kernel = "megakernel" if a.kernel == "auto" else a.kernel
    elif a.config == "C3":
        a.width, a.height, a.spp, a.level, a.material = 1920, 1080, 256, 6, 3
        a.kernel = "wavefront" if a.kernel == "auto" else a.kernel
    elif a.config in ("C4", "C5"):
        a.scene, a.level, a.material = "big", 8, 3
        a.kernel = "wavefront" if a.kernel == "auto" else a.kernel
        a.width, a.height, a.spp = (1920, 1080, 1024) if a.config == "C4" else (3840, 2160, 4096)
        if a.gpus == 1 and a.simulate_rank is None:
            a.simulate_rank = 3
    return a


GROUND_V = np.array([[-1000, -3, 1000, 0, 1, 0], [-1000, -3, -1000, 0, 1, 0], [1000, -3, -1000, 0, 1, 0], [1000, -3, 1000, 0, 1, 0]], np.float32)
GROUND_I = np.array([0, 1, 2, 2, 3, 0], np.uint32)
MAT_SPEC_DIFFUSE = dict(albedo=(0.8, 0.6, 0.2), specular=0.5)      # C2's material, index 4


def build_scene(P, args, mesh, aspect, renderer):
    """The shipped scene layout (ref: Main.cpp:777-819) around `mesh`: materials :779-782 (+ C2's), ground quad, two sphere
    lights, camera.  Meshes of a million triangles get their (bit-identical) SAH tree from the GPU builder."""
    s = P.Scene()
    for m in P.REFERENCE_MATERIALS:
        s.add_material(m)
    s.add_material(P.Material(**MAT_SPEC_DIFFUSE))
    s.add_mesh(mesh, args.material, P.BUILD_SAH_INTERVALS, device_builder=renderer if mesh.num_triangles > 400000 else None)
    s.add_mesh(P.Mesh.from_arrays(GROUND_V, GROUND_I), 1, P.BUILD_SAH_INTERVALS)
    for c in ((10.0, 10.0, 10.0), (-10.0, 10.0, -10.0)):
        s.add_light(s.add_sphere(c, 5.0, 2))
    if args.scene == "big":
        s.set_camera((0.0, 4.0, 30.0), (0.0, 0.0, -1.0), 60.0, aspect)
    else:
        s.set_camera((0, 0, 8), (0, 0, -1), 60.0, aspect)
    mode = {"advanced": P.MODE_ADVANCED, "brute": P.MODE_BRUTE_FORCE, "comparison": P.MODE_COMPARISON}[args.mode]
    s.set_settings(P.Settings(render_mode=mode))     # the reference's default settings (Main.cpp:228-235)
    return s


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
    for m in list(P.REFERENCE_MATERIALS) + [P.Material(**MAT_SPEC_DIFFUSE)]:
        o.add_material(m.albedo, m.specular, m.refractivity, m.absorption, m.ior, m.emissive, m.intensity, m.is_light)
    o.add_mesh(vertices, indices, args.material, O.BUILD_SAH_INTERVALS)
    o.add_mesh(GROUND_V, GROUND_I, 1, O.BUILD_SAH_INTERVALS)
    for c in ((10.0, 10.0, 10.0), (-10.0, 10.0, -10.0)):
        o.add_light(o.add_sphere(c, 5.0, 2))
    if args.scene == "big":
        o.set_camera((0.0, 4.0, 30.0), (0.0, 0.0, -1.0), 60.0, aspect)
    else:
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


def device_code_hash():
    """sha256 over the device sources the kernels are built from: the PMC figures in profiles/pmc_counts.json carry the hash of the
    code they were counted on, and a roofline fraction is only reported when it matches the code that just ran."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(REPO, "cpugpupathtracing_amd", "csrc", "device")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.hpp")) + glob.glob(os.path.join(d, "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


KERNEL_NAMES = {1: "megakernel", 2: "wf_trace", 3: "pt_persistent"}     # cgpt_stats.last_kernel -> the dominant kernel's name


def main():
    args = parse_args()
    torchrun = "WORLD_SIZE" in os.environ                      # launched by torch.distributed.run: one process per GPU
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if torchrun and args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus < 1 or args.gpus > 8:
        raise SystemExit("--gpus must be 1..8 (one node)")
    # in-process multi-device context (the C ABI's own multi-GPU host): N > 1 without a launcher, or the N = 1 rehearsals
    in_process = not torchrun and (args.gpus > 1 or args.force_collective or args.share_gpu)
    n_ranks = args.gpus if in_process else world

    import cpugpupathtracing_amd as P
    from cpugpupathtracing_amd import distributed as D

    torch = None
    dist = None
    if torchrun:
        import torch
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the render path")
        if args.rehearse_gloo:
            local_rank = 0
        torch.cuda.set_device(local_rank)
    collective = torchrun and (world > 1 or args.force_collective)
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
        if args.scene == "big":
            # ~1.3 M triangles: four times the stand-in's size, so the triangles stay above the reference's absolute determinant
            # epsilon (SURVEY A-9: at the stand-in's own size every level-8 triangle is rejected); tests/test_baseline_configs.py
            gen = P.Mesh.bumpy_icosphere(args.level, (0.0, 6.0, -30.0), (24.0, 10.0, 16.0), 0.15)
        else:
            gen = P.Mesh.dragon_standin(args.level)
        path = os.path.join(tmp, f"standin_l{args.level}_r{rank}.gltf")
        gen.save_gltf(path)
        mesh = P.Mesh.load_gltf(path)
    n_tris = mesh.num_triangles

    try:
        if in_process:
            # cgpt_ctx_create's own message is the whole failure (e.g. "device id 1 out of range (1 devices)" on a one-GPU box)
            if args.share_gpu:
                renderer = P.Renderer([0] * args.gpus, flags=P.CTX_GATHER_PEER_COPY)
            else:
                renderer = P.Renderer(list(range(args.gpus)), flags=P.CTX_FORCE_COLLECTIVE if args.gpus == 1 else 0)
        else:
            renderer = P.Renderer(local_rank)
    except P.DeviceError as e:
        raise SystemExit(f"bench.py --gpus {args.gpus}: {e}")
    scene = build_scene(P, args, mesh, aspect, renderer)
    renderer.upload(scene)
    if in_process and args.band_rows != 4:
        renderer.set_tuning(band_rows=args.band_rows)
    if args.pools:
        renderer.set_tuning(pools=args.pools)
    kernel = {"auto": P.KERNEL_AUTO, "policy": P.KERNEL_AUTO, "megakernel": P.KERNEL_MEGAKERNEL, "wavefront": P.KERNEL_WAVEFRONT, "persistent": P.KERNEL_PERSISTENT}[args.kernel]
    # N > 1: 4-row bands dealt round-robin over the ranks, so every GPU gets the same mix of sky, mesh and ground rows (the multi-device
    # context does this tiling itself)
    interleave = (args.band_rows, world, rank) if torchrun and world > 1 else None
    n_rows = len(D.interleaved_rows(args.height, rank, world, args.band_rows)) if torchrun and world > 1 else args.height
    if args.simulate_rank is not None and n_ranks == 1 and not in_process:
        interleave = (args.band_rows, args.simulate_world, args.simulate_rank)
        n_rows = len(D.interleaved_rows(args.height, args.simulate_rank, args.simulate_world, args.band_rows))
    gather = None
    if collective:
        gather = D.FramebufferGather(args.width, args.height, rank, world, local_rank, band_rows=args.band_rows,
                                     device="cpu" if args.rehearse_gloo else None)

    def sync():
        renderer.synchronize()
        if torch is not None:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def step(counters=False):
        renderer.reset_accumulator()
        renderer.render(args.width, args.height, args.spp, seed=args.seed, interleave=interleave, kernel=kernel, counters=counters)
        if in_process:
            renderer.accumulator_device_ptr()        # the gathered float4 frame on GPU 0: forces the one exchange of the step
        elif gather is not None:
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

    total_rays = float(st.traced_rays)
    if dist is not None:
        red_dev = "cpu" if args.rehearse_gloo else "cuda"
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        rays = torch.tensor([total_rays], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
        elapsed = float(t.item())
        total_rays = float(rays.item())

    rehearsal = None
    if rank == 0 and ((in_process and (args.share_gpu or args.force_collective)) or (torchrun and (args.rehearse_gloo and world > 1 or args.force_collective and world == 1))):
        # rehearsal check: the gathered, re-ordered framebuffer equals a single-context render of the whole frame
        gathered = renderer.accumulator() if in_process else step.full.cpu().numpy()
        one = P.Renderer(0) if in_process else renderer
        if in_process:
            one.upload(scene)
        one.reset_accumulator()
        one.render(args.width, args.height, args.spp, seed=args.seed, kernel=kernel)
        rehearsal = bool(np.array_equal(one.accumulator().view(np.uint32), gathered.view(np.uint32)))
        if in_process:
            one.close()
        print(f"[rehearsal] gathered framebuffer identical to the single-GPU render: {rehearsal}", file=sys.stderr, flush=True)
        if not rehearsal:
            raise SystemExit("rehearsal mismatch")

    # ---- roofline pass (untimed, N = 1): the dominant kernel with the chip to itself, and the measured issue roof ----
    dominant = KERNEL_NAMES.get(st.last_kernel, "wf_trace")     # which kernel the timed steps ran: reported by the library (AUTO resolved)
    wavefront, persistent = dominant == "wf_trace", dominant == "pt_persistent"
    excl = None
    peak_rate = None
    if not args.no_roofline_pass and n_ranks == 1:
        if wavefront:
            renderer.set_tuning(pools=1)
            step()                                   # re-sizes the pools for one batch in flight
        if persistent:
            renderer.set_tuning(pt_streams=1)
        renderer.reset_stats()
        sync()
        step()
        sync()
        excl = renderer.stats()
        waves = max(1, min(8, excl.dominant_waves_per_simd))
        # a SIMD issues one wave64 VALU instruction every 2 cycles only while two waves alternate (one wave alone: every 4), so an odd
        # number of resident waves measures between the two (profiles/r02/issue_rate_table.txt): the roof is taken at the next even count
        peak_waves = min(8, waves + (waves & 1))
        peak_rate, _ = renderer.measure_issue_rate(kind=0, waves_per_simd=peak_waves, iters=40000)
        if args.issue_table and rank == 0:
            names = ["v_mul_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_rcp_f32", "3 v_mul : 1 v_pk_mul interleaved", "48 v_mul + 16 v_pk_mul grouped",
                     "1 v_mul : 1 v_pk_mul alternating", "v_cndmask_b32 (vcc)", "v_mul_lo_u32", "v_cndmask_b32_e64 (sgpr pair)",
                     "v_cmp_lt_f32 + v_cndmask_b32 pairs", "v_add_u32", "v_min3_f32"]
            print("[issue rate] Gwave-inst/s over the chip, by waves per SIMD (1..8)", file=sys.stderr)
            for kind, nm in enumerate(names):
                row = [renderer.measure_issue_rate(kind=kind, waves_per_simd=w, iters=60000)[0] / 1e9 for w in range(1, 9)]
                print(f"[issue rate] {nm:34s} " + " ".join(f"{v:8.1f}" for v in row), file=sys.stderr, flush=True)

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        launches_per_step = st.dominant_launches / max(1, args.steps)
        bytes_per_ray = b_alg / max(1, rays_per_step_local)
        # PMC-derived per-step figures for exactly this workload (scripts/gpu_roofline_pmc.sh writes them)
        pmc = None
        key = f"{args.scene}_{args.width}x{args.height}x{args.spp}_l{args.level}_m{args.material}_{dominant}_rows{n_rows}"
        ppath = os.path.join(REPO, "profiles", "pmc_counts.json")
        if os.path.exists(ppath):
            try:
                pmc = json.load(open(ppath)).get(key)
            except Exception:
                pmc = None
        code_hash = device_code_hash()
        roof = {"bound": "valu_issue", "kernel": dominant, "achieved": None, "peak": None, "unit": "Gwave-inst/s", "frac": None,
                "traffic": None, "hbm_frac": None, "mem_return_frac": None, "pmc_key": key, "code_hash": code_hash,
                "algorithmic_bytes_per_ray": round(bytes_per_ray, 2),
                "algorithmic_gbs": round(b_alg / (st.kernel_ms / max(1, args.steps) * 1e-3) / 1e9, 2),
                "launches_per_step": round(launches_per_step, 1),
                "render_ms_per_step": round(st.kernel_ms / max(1, args.steps), 3)}
        if excl is not None:
            k_ms = excl.dominant_ms                  # sum over the launches of ONE step, each alone on the chip
            k_n = max(1, excl.dominant_launches)
            roof.update({"peak": round(peak_rate / 1e9, 2), "peak_source": f"cgpt_measure_issue_rate: v_mul_f32 streams, {peak_waves} waves/SIMD (kernel: {waves}), 256 CUs, this run",
                         "kernel_ms_per_step": round(k_ms, 3), "kernel_ms_per_launch": round(k_ms / k_n, 4),
                         "exclusive_pass_ms_per_step": round(excl.kernel_ms, 3), "waves_per_simd": waves,
                         "timing": ("hipEvents around every wf_trace launch in a single-pool pass (one batch in flight)" if wavefront
                                    else f"hipEvents around the {dominant} launch(es), one batch in flight")})
            if wavefront and excl.dominant_round0_launches:
                # Two ray populations in one number (the reference does not jitter, SURVEY A-14: the 64 primary rays of a wave are identical and
                # walk the tree in lockstep; every one of them is an executed, counted IntersectScene call): the trace kernel's own rate on each
                r0 = float(args.width) * n_rows * args.spp      # primary rays = paths
                later = max(0.0, float(excl.traced_rays) - r0)
                r0_ms, later_ms = excl.dominant_round0_ms, max(1e-9, k_ms - excl.dominant_round0_ms)
                roof.update({"rays_round0": int(r0), "rays_later": int(later),
                             "trace_ms_round0": round(r0_ms, 3), "trace_ms_later": round(later_ms, 3),
                             "trace_grays_round0": round(r0 / (r0_ms * 1e-3) / 1e9, 2) if r0_ms > 0 else None,
                             "trace_grays_later": round(later / (later_ms * 1e-3) / 1e9, 2)})
            if pmc and pmc.get("code_hash") != code_hash:
                roof.update({"pmc_stale": True, "pmc_stale_reason": f"profiles/pmc_counts.json[{key}] was counted on device code {pmc.get('code_hash')}, "
                                                                    f"this run built {code_hash}: re-run scripts/gpu_roofline_pmc.sh"})
            elif pmc:
                insts = float(pmc["dominant_valu_wave_insts_per_step"])
                hbm = float(pmc["dominant_hbm_bytes_per_step"])
                roof.update({"achieved": round(insts / (k_ms * 1e-3) / 1e9, 2),
                             "frac": round(insts / (k_ms * 1e-3) / peak_rate, 4),
                             "valu_wave_insts_per_launch": int(insts / k_n),
                             "active_lane_frac": pmc.get("dominant_active_lane_frac"),
                             "traffic": int(hbm / k_n), "hbm_frac": round(hbm / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                             # the vector-memory return path (texture-data unit busy share of the kernel's time): the nearest roof of the
                             # later trace rounds -- a returned dword costs a 64-lane register write whatever the lane count
                             "mem_return_frac": pmc.get("dominant_td_busy_frac"),
                             "mem_return_frac_later_rounds": pmc.get("later_rounds_td_busy_frac"),
                             # every kernel of the step against the same roof, over the production (overlapped) step time
                             "pipeline_frac": round(float(pmc["all_valu_wave_insts_per_step"]) / (ms_per_step * 1e-3) / peak_rate, 4),
                             "pipeline_hbm_frac": round(float(pmc["all_hbm_bytes_per_step"]) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                             "pmc_source": pmc.get("source")})
        if n_ranks > 1:
            roof["note"] = "roofline of the dominant kernel is measured at N = 1 (python bench.py)"
        if in_process:
            parallelism = (f"in-process multi-device context: {args.band_rows}-row bands interleaved over {n_ranks} rank(s), "
                           + ("all on GPU 0, peer-copy gather (rehearsal, not a measurement)" if args.share_gpu
                              else f"one grouped RCCL send/recv exchange of the float4 bands per step ({st.rccl_ranks} RCCL ranks)"))
        elif world > 1:
            parallelism = f"one process per GPU: {args.band_rows}-row bands interleaved over {world} GPUs + 1 RCCL gather/step (torch.distributed)"
        else:
            parallelism = "1 GPU"
        out = {
            "metric": f"Mrays/sec @ {args.width}x{args.height}, {args.spp} spp (dragon glTF stand-in)",
            "value": round(total_rays / elapsed / 1e6, 3),
            "unit": "Mrays/s",
            "n_gpus": n_ranks,
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
                "workload": (f"{args.config + ': ' if args.config else ''}"
                             + (f"bumpy icosphere level {args.level} ({n_tris} tris, 4x the stand-in's size)" if args.scene == "big"
                                else f"dragon stand-in (bumpy icosphere level {args.level}, {n_tris} tris)")
                             + f", SAH-intervals BVH, via glTF, in the reference scene layout (Main.cpp:777-819), material {args.material}, "
                             f"{args.width}x{args.height}, {args.spp} spp, render_mode {args.mode}, default settings (NEE, RR, cosine, max depth 5)"
                             + (f"; rank {args.simulate_rank} of {args.simulate_world}'s interleaved bands only" if interleave is not None and n_ranks == 1 else "")),
                "kernel": args.kernel, "kernel_ran": dominant, "rays_per_step": int(total_rays / args.steps), "rows_per_gpu": n_rows if not in_process else None,
                "parallelism": parallelism,
            },
            "roofline": roof,
        }
        if in_process:
            out["config"].update({
                "host": "in_process", "rccl_ranks": st.rccl_ranks, "gathers_per_step": round(st.gathers / max(1, args.steps), 2),
                "gather_ms": round(st.gather_ms / max(1, st.gathers), 4),                 # per exchange: RCCL send/recv (or peer copies) + reorder
                "device_ms": [round(st.device_ms[r] / max(1, args.steps), 3) for r in range(n_ranks)],     # render time per step on every GPU
                "gathered_image_checked": rehearsal})
        elif torchrun:
            out["config"]["host"] = "torch.distributed.run"
        if args.cpu_seconds > 0 and n_ranks == 1:
            sys.path.insert(0, os.path.join(REPO, "oracle"))
            out["cpu_baseline"] = cpu_baseline(args, mesh.vertices, mesh.indices, aspect)
        print(json.dumps(out), flush=True)

    renderer.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
