import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests skip cleanly where there is no GPU (they are selected with -m gpu on the box)."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """Build libsrk_gfx950.so when the tree has none yet (fresh checkout: built artefacts are git-ignored).  hipcc
    cross-compiles without a GPU; where there is no hipcc either, the tests that need the library fail loudly."""
    lib = os.path.join(ROOT, "sr-pytorch-lightning_amd", "libsrk_gfx950.so")
    if not os.path.exists(lib) and (os.path.exists("/opt/rocm/bin/hipcc") or __import__("shutil").which("hipcc")):
        import __graft_entry__
        __graft_entry__.build()
    yield
