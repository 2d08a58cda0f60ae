import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(params=[1, 0], ids=["split_bf16", "fp32_mfma"])
def tap_kernel(request):
    """Runs a test under both kernels of the fp32 tap-GEMM (option split_bf16; tests/helpers.py: tap_gemm_kernel)."""
    from tests.helpers import tap_gemm_kernel
    with tap_gemm_kernel(request.param) as k:
        yield k
