import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")


def _gpu_count():
    try:
        from fitsnap_amd import _capi

        return _capi.device_count()
    except Exception:
        return 0


def pytest_collection_modifyitems(config, items):
    """A plain `pytest` on a box without a gfx950 device skips the gpu-marked tests instead of failing them."""
    if _gpu_count() > 0:
        return
    skip = pytest.mark.skip(reason="no gfx950 GPU visible (run with -m gpu on the GPU box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def ta():
    """Golden Ta matrices committed by the reference
    (examples/Ta_Linear_JCP2014/20May21_Standard/{Descriptors,Truth-Ref,Weights}.npy)."""
    d = np.load(os.path.join(GOLDEN, "ta_abw.npz"))
    return np.ascontiguousarray(d["A"]), np.ascontiguousarray(d["b"]), np.ascontiguousarray(d["w"])


@pytest.fixture(scope="session")
def ta_fits():
    """Outputs of the reference's own SVD / RIDGE classes on the golden matrices
    (tests/golden/make_golden.py)."""
    return dict(np.load(os.path.join(GOLDEN, "ta_reference_fits.npz")))


def relerr(x, ref):
    x, ref = np.asarray(x, dtype=float), np.asarray(ref, dtype=float)
    return float(np.max(np.abs(x - ref)) / np.max(np.abs(ref)))


def maxrel(x, ref):
    x, ref = np.asarray(x, dtype=float).ravel(), np.asarray(ref, dtype=float).ravel()
    return float(np.max(np.abs(x - ref) / np.maximum(np.abs(ref), 1e-300)))
