import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


HAS_GPU = _has_gpu()


@pytest.fixture(scope="session", autouse=True)
def _build_native():
    """Oracle + scene generator are CPU builds (seconds); the HIP library must already be built
    in-tree (``python -m balm_amd.build`` / ``__graft_entry__.build()``), it is never a fallback."""
    from oracle import orc
    from balm_amd import scene
    orc.build()
    from oracle import assoc_host
    assoc_host.build()
    scene.build()
    yield


def pytest_collection_modifyitems(config, items):
    if HAS_GPU:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
