import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

REFERENCE_DIR = "/root/reference"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")


def _gpu_present() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    if _gpu_present():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def reference_assets():
    """Cube/Duck glTF assets of the reference checkout (absent on the GPU box)."""
    p = os.path.join(REFERENCE_DIR, "Assets", "Models")
    if not os.path.isdir(p):
        pytest.skip("/root/reference is not present here")
    return p


@pytest.fixture(scope="session", autouse=True)
def _built_libraries():
    """Build the product library and the oracle if they are stale (both are plain compiler invocations)."""
    from cpugpupathtracing_amd import build as product_build
    import oracle
    product_build.build()
    oracle.build()
