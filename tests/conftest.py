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


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _options_back_to_their_defaults(request):
    """a GPU test may force a path with ops.set_options(...): whatever it set is gone before the next test (the options are process-wide)"""
    yield
    if "gpu" in request.keywords and _has_gpu():
        from datafusion_amd import _lib
        if _lib._lib is not None:
            _lib._lib.dfgpu_set_option(None, None)
