import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_collection_modifyitems(config, items):
    """The -m gpu tests need a HIP device: skip (not fail) them where there is none."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no HIP device in this environment (run on the GPU box: pytest -m gpu)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _expected_library_switches():
    """A test run started with GFL_EXPECT_EWA_MFMA=1 (tests/test_gpu_primitives.py starts one) must really be running
    the matrix-core contraction: the library reads GFL_EWA_MFMA once per process."""
    if os.environ.get("GFL_EXPECT_EWA_MFMA") == "1":
        from gflow_amd import _lib
        assert _lib.load().gfl_ewa_on_mfma() == 1, "GFL_EWA_MFMA=1 did not reach the library"
    yield
