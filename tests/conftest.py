"""pytest configuration: registers the ``gpu`` marker and puts the repo root on sys.path.

``-m "not gpu"``: oracle vs the reference's golden vectors, host logic, and the C-ABI library's
exported symbols (no GPU needed).  ``-m gpu``: parity tests proper, through the C ABI on an MI355X.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no binaries (they are git-ignored): build the HIP library (cross-compiles without a
    GPU), the host library and the C oracle once, exactly as ``__graft_entry__.build()`` does."""
    needed = [os.path.join(ROOT, "whvi_amd", "libwhvi_hip.so"), os.path.join(ROOT, "whvi_amd", "libwhvi_cpu.so")]
    if all(os.path.exists(p) for p in needed):
        return
    import __graft_entry__
    __graft_entry__.build()


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def hip_lib():
    """The native library must be there on a GPU box -- fail loudly, never fall back."""
    from whvi_amd import _hip
    return _hip.lib()


@pytest.fixture(params=["auto", "faithful"])
def dataflow(request):
    """Run a test on both evaluation routes of the square weight matrix (whvi_amd/weights.py): "auto" = the one-launch
    diagonal application on the GPU and the one-launch products of the stacked / column layers (the shipped defaults),
    "faithful" = weight construction through the FWHT kernels + dense GEMMs everywhere, the reference's dataflow op for op
    (src/weights.py:87-93,179-180,250-251)."""
    from whvi_amd.weights import WHVIColumnMatrix, WHVISquarePow2Matrix, WHVIStackedMatrix
    # (not through ``monkeypatch``: several of these tests call monkeypatch.undo() half way to drop a torch.randn replay)
    before = (WHVISquarePow2Matrix.default_exploit_diagonal, WHVIStackedMatrix.hip_apply, WHVIColumnMatrix.hip_apply)
    faithful = request.param != "auto"
    WHVISquarePow2Matrix.default_exploit_diagonal = False if faithful else "auto"
    WHVIStackedMatrix.hip_apply = WHVIColumnMatrix.hip_apply = not faithful        # torch.matmul instead of the one-launch products
    yield request.param
    WHVISquarePow2Matrix.default_exploit_diagonal, WHVIStackedMatrix.hip_apply, WHVIColumnMatrix.hip_apply = before
