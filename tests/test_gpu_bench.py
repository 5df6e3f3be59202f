"""bench.py's contract on a small workload, and the RCCL collective path of N > 1 exercised at world size 1.

The driver runs bench.py at N = 1, 2, 4, 8; a one-GPU box cannot host two RCCL ranks, so the N > 1 plumbing is covered in
two halves: tiling + gather + reorder on gloo (tests/test_distributed.py, and bench.py --rehearse-gloo), and here the
nccl (= RCCL) process group, the zero-copy device-pointer gather and the row reorder with a single rank."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--width", "256", "--height", "144", "--spp", "4", "--level", "3", "--steps", "1", "--warmup", "1"]


def run_bench(extra, env_extra=None):
    env = dict(os.environ)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + SMALL + extra, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0]), p.stderr


def test_bench_line_contract():
    d, _ = run_bench(["--cpu-seconds", "0.5", "--cpu-threads", "2"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "Mrays/s" and d["n_gpus"] == 1 and d["vs_baseline"] is None and d["value"] > 0
    r = d["roofline"]
    # the roof is vector-instruction issue, measured in the same process; the PMC-derived numerator exists only for workloads
    # that were profiled (profiles/pmc_counts.json), so `achieved` / `frac` may be null on this small one
    assert r["bound"] == "valu_issue" and r["unit"] == "Gwave-inst/s" and 300.0 < r["peak"] < 2000.0
    assert r["kernel_ms_per_step"] > 0 and r["kernel_ms_per_step"] <= r["exclusive_pass_ms_per_step"] * 1.001
    assert 1 <= r["waves_per_simd"] <= 8 and r["algorithmic_gbs"] > 0
    if r["frac"] is not None:
        assert 0 < r["frac"] <= 1.0 and 0 <= r["hbm_frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 2 and c["value"] > 0 and c["unit"] == "Mrays/s"
    assert "workload" in d["config"] and "model" not in d["config"]


def _run_raw(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + SMALL + ["--cpu-seconds", "0"] + extra, capture_output=True, text=True, timeout=600, env=env)


def test_gpus_n_runs_without_a_launcher_through_the_multi_device_context():
    """`python bench.py --gpus N` as a plain command (no torchrun, WORLD_SIZE unset) goes through the C ABI's multi-device context
    (csrc/device/multi_gpu.hip).  On a one-GPU box --gpus 2 can only fail -- with cgpt_ctx_create's own one-line message; with two
    or more GPUs it must run and report per-device times and the RCCL exchange."""
    import torch
    p = _run_raw(["--gpus", "2"])
    if torch.cuda.device_count() >= 2:
        assert p.returncode == 0, p.stderr[-2000:]
        d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
        assert d["n_gpus"] == 2 and d["config"]["rccl_ranks"] == 2 and len(d["config"]["device_ms"]) == 2 and d["config"]["gather_ms"] > 0
    else:
        assert p.returncode != 0
        err = [l for l in p.stderr.splitlines() if l.strip()]
        assert len(err) == 1 and "device id 1 out of range (1 devices)" in err[0], p.stderr[-2000:]
        assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_gpus_2_share_gpu_rehearsal_checks_the_gathered_image():
    p = _run_raw(["--gpus", "2", "--share-gpu"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert "identical to the single-GPU render: True" in p.stderr
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    c = d["config"]
    assert d["n_gpus"] == 2 and c["host"] == "in_process" and c["rccl_ranks"] == 0 and c["gathered_image_checked"] is True
    assert len(c["device_ms"]) == 2 and all(t > 0 for t in c["device_ms"]) and c["gathers_per_step"] == 1 and c["gather_ms"] > 0
    assert d["value"] > 0 and d["scaling"] == "strong"


def test_in_process_rccl_exchange_with_one_rank():
    p = _run_raw(["--gpus", "1", "--force-collective"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert "identical to the single-GPU render: True" in p.stderr
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["config"]["rccl_ranks"] == 1 and d["config"]["gather_ms"] > 0 and d["config"]["gathers_per_step"] == 1


def test_rccl_collective_path_at_world_size_one():
    port = 29600 + os.getpid() % 300
    d, err = run_bench(["--cpu-seconds", "0", "--force-collective"], {"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": "0",
                                                                       "LOCAL_RANK": "0", "WORLD_SIZE": "1"})
    assert "identical to the single-GPU render: True" in err, err[-2000:]
    assert d["n_gpus"] == 1 and d["value"] > 0


_RENDER_SNIPPET = """
import sys, numpy as np
sys.path.insert(0, {repo!r})
import cpugpupathtracing_amd as P
mesh = P.Mesh.dragon_standin(3)
r = P.Renderer(0)
r.upload(P.Scene.reference_layout(mesh, 3, 2.0, P.BUILD_SAH_INTERVALS))
r.render(512, 256, 64, seed=7, kernel=P.KERNEL_WAVEFRONT)
np.save({out!r}, r.accumulator())
print(r.stats().traced_rays)
"""


def test_batch_sizing_does_not_change_the_image(tmp_path):
    """The wavefront path sizes its sample batches against the free HBM; whatever it picks (default: one or two big batches;
    a 1-GiB budget with 4-sample batches in up to 8 pools; a single pool), samples are accumulated in order, so the image
    and the ray count are the same bit for bit."""
    import numpy as np
    outs, rays = [], []
    for k, env in enumerate(({}, {"CGPT_WF_BUDGET_GIB": "1", "CGPT_WF_MAX_BATCH": "4"}, {"CGPT_WF_POOLS": "1", "CGPT_WF_BATCH": "5"})):
        out = str(tmp_path / f"acc{k}.npy")
        e = dict(os.environ); e.update(env)
        p = subprocess.run([sys.executable, "-c", _RENDER_SNIPPET.format(repo=REPO, out=out)], capture_output=True, text=True, timeout=300, env=e)
        assert p.returncode == 0, p.stderr[-2000:]
        outs.append(np.load(out)); rays.append(int(p.stdout.split()[-1]))
    assert rays[0] == rays[1] == rays[2] and rays[0] > 512 * 256 * 64
    assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))
    assert np.array_equal(outs[0].view(np.uint32), outs[2].view(np.uint32))
