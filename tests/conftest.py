import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def ref_modules():
    """oracle/_ref pybind modules (the reference's own sources compiled for the host)."""
    from oracle import build_ref

    def get(name):
        if not build_ref.so_path(name).exists():
            if build_ref.available():
                build_ref.build(verbose=False)
            else:
                pytest.skip(f"oracle/_ref/{name}.so not built and /root/reference absent")
        return build_ref.load(name)
    return get
