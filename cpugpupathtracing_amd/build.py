"""Builds libcpugpupt.so (host mirror + HIP kernels + C ABI) in-tree with hipcc for gfx950.

    python -m cpugpupathtracing_amd.build [--force]

hipcc cross-compiles without a GPU.  -ffp-contract=off on host AND device code: parity with the reference
needs identical float operation order with no FMA contraction.
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys

from ._paths import CSRC, LIB_DIR, LIB_PATH, PKG_DIR, REPO_DIR  # noqa: F401  (re-exported)

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
COMMON_FLAGS = ["-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wall", "-Wextra", "-Wno-unused-parameter",
                "-I" + os.path.join(REPO_DIR, "include"), "-I" + os.path.join(CSRC, "host"), "-I" + os.path.join(CSRC, "device")]
# -fno-slp-vectorize: left alone, the SLP vectoriser turns the scalar triangle / shading arithmetic into packed-f32 instructions fed by
# register shuffles (v_mov, v_pk_mov): more instructions and 6-12 more VGPRs per kernel; measured 3.6 % slower end to end.  The
# packed math that pays (both children of a BVH node at once) is written explicitly (rt_device.hpp: slab_products).
DEVICE_FLAGS = ["--offload-arch=gfx950", "-fno-slp-vectorize"] + os.environ.get("CGPT_EXTRA_HIPCC_FLAGS", "").split()


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "device", "*.hip"))) + sorted(glob.glob(os.path.join(CSRC, "host", "*.cpp")))


def _deps():
    d = sources()
    for pat in ("device/*.h", "device/*.hpp", "host/*.h"):
        d += glob.glob(os.path.join(CSRC, pat))
    d += glob.glob(os.path.join(REPO_DIR, "include", "*.h"))
    d.append(os.path.abspath(__file__))
    return d


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(p) > t for p in _deps())


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    # objects of a variant build (CGPT_LIB_PATH=.../libcpugpupt_<tag>.so with CGPT_EXTRA_HIPCC_FLAGS) stay apart from the product's
    tag = os.path.splitext(os.path.basename(LIB_PATH))[0].replace("libcpugpupt", "")
    obj_dir = os.path.join(LIB_DIR, "obj" + tag)
    os.makedirs(obj_dir, exist_ok=True)
    objs = []
    procs = []
    for src in sources():
        obj = os.path.join(obj_dir, os.path.basename(src) + ".o")
        objs.append(obj)
        cmd = [HIPCC] + COMMON_FLAGS + (DEVICE_FLAGS if src.endswith(".hip") else []) + ["-c", src, "-o", obj]
        if src.endswith(".cpp"):
            cmd.insert(1, "-x"); cmd.insert(2, "c++")     # host-only translation units: no device pass
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"--- {src}\n{out}\n")
        elif verbose and out.strip():
            print(out)
    if failed:
        raise RuntimeError("hipcc failed; see messages above")
    # librccl: the multi-device context's framebuffer gather (csrc/device/multi_gpu.hip)
    rocm_lib = os.path.join(os.path.dirname(os.path.dirname(HIPCC)), "lib")
    link = [HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB_PATH] + objs + ["-L" + rocm_lib, "-lrccl", "-lpthread", "-Wl,-rpath," + rocm_lib]
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    return LIB_PATH


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose="-v" in sys.argv)
    print(path)
