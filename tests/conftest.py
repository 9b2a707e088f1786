import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture
def diag_lib():
    """For the duration of the test the library is the diagnostics build (libacez_diag.so = the product's sources with -DACEZ_DIAG): the
    only build in which the ACEZ_* ablation switches, the measured-and-rejected kernels (chain_kernel, headfwd_kernel, 128-row GEMM tiles,
    wgrad256) and the fault-injection hooks exist. Every test that sets such a switch asks for this fixture."""
    from acezero_amd import _native as N
    with N.diag_library() as lib:
        yield lib


def pytest_collection_modifyitems(config, items):
    """Tests marked `gpu` need an MI355X: on a machine without one they are skipped (not failed), so a plain `pytest tests` is green
    in the build container; the product itself still raises without a GPU (tests/test_abi.py::test_no_silent_cpu_fallback)."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a GPU (MI355X): run with `-m gpu` on the GPU box")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
