import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

PKG_NAME = "tcn-transformer-gan-hybrid-model-for-predicting-streamflow-in-ungauged-basins_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_pkg():
    return importlib.import_module(PKG_NAME)


@pytest.fixture(scope="session")
def pkg():
    return load_pkg()


@pytest.fixture(scope="session")
def oracle():
    return importlib.import_module("oracle.streamflow_oracle")


def golden(name):
    return np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"), allow_pickle=False)


def _np(a):
    import torch
    if isinstance(a, torch.Tensor):
        a = a.detach().float().cpu().numpy() if a.dtype != torch.float64 else a.detach().cpu().numpy()
    return np.asarray(a, dtype=np.float64)


def close(a, b, rtol=1e-5):
    """max |a-b| / max |b| <= rtol."""
    a, b = _np(a), _np(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    if b.size == 0:
        return
    scale = max(np.abs(b).max(), 1e-30)
    err = np.abs(a - b).max() / scale
    assert np.isfinite(a).all(), "non-finite values"
    assert err <= rtol, f"max rel-to-max err {err:.3e} > {rtol}"


def close_elementwise(a, b, rtol=1e-4, atol_frac=2e-6):
    """Element-wise |a-b| <= rtol |b| + atol_frac max|b| (numpy.allclose with the absolute term tied
    to the tensor's scale): unlike close(), small elements are held to their own size, down to the
    cancellation noise of an f32 sum (a few 1e-6 of the largest element)."""
    a, b = _np(a), _np(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    if b.size == 0:
        return
    bound = rtol * np.abs(b) + atol_frac * max(np.abs(b).max(), 1e-30)
    excess = np.abs(a - b) - bound
    assert excess.max() <= 0, f"{int((excess > 0).sum())}/{b.size} elements beyond rtol {rtol} + {atol_frac} of max"


def close_robust(a, b, rtol=1e-5, frac=0.01, loose=0.05):
    """As close(), but tolerates a small fraction of outliers (up to `loose`): a ReLU /
    max-pool decision on a pre-activation within rounding of zero flips between two
    correct implementations and changes the gradient in a small neighbourhood."""
    a, b = _np(a), _np(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    if b.size == 0:
        return
    assert np.isfinite(a).all(), "non-finite values"
    scale = max(np.abs(b).max(), 1e-30)
    e = np.abs(a - b) / scale
    nbad = int((e > rtol).sum())
    assert nbad <= max(frac * e.size, 0), f"{nbad}/{e.size} elements beyond {rtol} (max {e.max():.3e})"
    assert e.max() <= loose, f"max rel-to-max err {e.max():.3e} > loose bound {loose}"
