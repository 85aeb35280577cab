import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip():
    """the product library; fails loudly when it is not built or no GPU is visible"""
    from jasper_amd import _lib
    L = _lib.lib()
    import ctypes as C
    n = C.c_int(0)
    _lib.check(L.jasper_device_count(C.byref(n)))
    assert n.value >= 1, "no HIP device visible"
    return L


@pytest.fixture(autouse=True)
def _give_cached_gpu_memory_back(request):
    """a GPU test that used torch tensors leaves them in torch's caching allocator; the CLI tests run OTHER processes (up to
    two ranks sharing the one GPU, tables of tens of GB each) that need that memory"""
    yield
    if request.node.get_closest_marker("gpu") is not None and "torch" in sys.modules:
        import torch
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
