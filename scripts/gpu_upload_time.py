#!/usr/bin/env python3
"""On the GPU box: what a host pays outside cgpt_render for a scene of the stand-in mesh at a given level: Scene construction with
the tree built on the GPU, flatten (the reference-layout arrays), cgpt_scene_upload (validation + device re-layout).
usage: python scripts/gpu_upload_time.py [levels ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cpugpupathtracing_amd as P

levels = [int(a) for a in sys.argv[1:]] or [6, 7, 8]
r = P.Renderer(0)
for level in levels:
    mesh = P.Mesh.dragon_standin(level) if level < 8 else P.Mesh.bumpy_icosphere(8, (0.0, 6.0, -30.0), (24.0, 10.0, 16.0), 0.15)
    t0 = time.perf_counter()
    s = P.Scene()
    for m in P.REFERENCE_MATERIALS:
        s.add_material(m)
    s.add_mesh(mesh, 3, P.BUILD_SAH_INTERVALS, device_builder=r)
    s.add_light(s.add_sphere((10.0, 10.0, 10.0), 5.0, 2))
    s.set_camera((0, 0, 8), (0, 0, -1), 60.0, 1.0)
    t1 = time.perf_counter()
    s.flatten()
    t2 = time.perf_counter()
    r.upload(s)
    t3 = time.perf_counter()
    r.upload(s)
    t4 = time.perf_counter()
    r.render(256, 256, 1)
    print(f"level {level}: {mesh.num_triangles} triangles: scene + GPU tree {1e3 * (t1 - t0):8.1f} ms, flatten {1e3 * (t2 - t1):7.1f} ms, "
          f"upload {1e3 * (t3 - t2):8.1f} ms (again: {1e3 * (t4 - t3):8.1f} ms)", flush=True)
