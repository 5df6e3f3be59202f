#!/usr/bin/env python3
"""On the GPU box: BVH::Build (SAH split intervals, ref: Source/BVH.cpp:11-45,204-366) of the stand-in mesh on the host (csrc/host/mesh_bvh.cpp,
one thread, as the reference) and on the GPU (csrc/device/bvh_build.hip, same tree bit for bit: tests/test_gpu_bvh_build.py).
usage: python scripts/gpu_bvh_build_time.py [levels ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cpugpupathtracing_amd as P

levels = [int(a) for a in sys.argv[1:]] or [5, 6, 7, 8]
r = P.Renderer(0)
for level in levels:
    mesh = P.Mesh.dragon_standin(level) if level < 8 else P.Mesh.bumpy_icosphere(8, (0.0, 6.0, -30.0), (24.0, 10.0, 16.0), 0.15)   # bench.py's 1.31 M-triangle scene
    n_tris = len(mesh.indices) // 3
    times = {}
    for name, builder in (("gpu (first call)", r), ("gpu", r), ("host", None)):
        s = P.Scene()
        s.add_material(P.Material())
        t0 = time.perf_counter()
        s.add_mesh(mesh, 0, P.BUILD_SAH_INTERVALS, device_builder=builder)
        times[name] = time.perf_counter() - t0
        info = s.bvh_info(0)
    print(f"level {level}: {n_tris} triangles, depth {info.max_depth}: host {times['host'] * 1e3:9.1f} ms, gpu {times['gpu'] * 1e3:8.1f} ms "
          f"(first call {times['gpu (first call)'] * 1e3:8.1f} ms), {times['host'] / times['gpu']:.1f}x", flush=True)
