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


# Biases of convolutions that feed a BatchNorm have an analytically zero gradient (the mean
# subtraction removes them); what reaches Adam is rounding noise, which Adam's normalisation turns
# into steps of +-lr with signs no two implementations share.  The key third of an attention
# in_proj_bias is the same case (softmax is invariant to a shift of all its logits).
NOISE_DRIVEN = ("model.2.0.bias", "model.3.0.bias", "model.4.0.bias")
PARTLY_NOISE_DRIVEN = ("self_attn.in_proj_bias",)


def check_update_projection(after, before, keys, ref, tol_proj=2e-2, tol_norm=2e-3, skip=NOISE_DRIVEN):
    """Signed projections of the parameter updates (goldgen.update_projection) against the golden
    ones, both relative to the reference update's L2 norm: the norms must agree to tol_norm, the
    projections to tol_proj (0.1 for tensors a third of whose entries are noise driven; None: norms only).  A
    missing, doubled or sign-flipped step on a tensor is an error of order 1 in both; a learning
    rate, bias correction or gradient scale off by a per cent shows in the norm.  What keeps the
    projection from 1e-3 even between two correct f32 implementations is Adam itself: an entry
    whose gradient is within rounding of zero still moves by about lr, in the direction of the
    noise."""
    import goldgen as G
    keys = [str(k) for k in keys]
    got = G.update_projection(after, before, keys)
    bad = []
    for k, want in zip(keys, ref):
        if k in skip:
            continue
        scale = max(float(want[1]), 1e-30)
        tp = 0.1 if k.endswith(PARTLY_NOISE_DRIVEN) else tol_proj
        e_proj, e_norm = abs(got[k][0] - want[0]) / scale, abs(got[k][1] - want[1]) / scale
        if not ((tol_proj is None or e_proj <= tp) and e_norm <= tol_norm):
            bad.append(f"{k}: projection off by {e_proj:.2e}, norm by {e_norm:.2e} of the update norm {scale:.3e}")
    assert not bad, "\n".join(bad)


def check_update_signs(after, before, keys, ref_bits, min_agree=0.99, skip=NOISE_DRIVEN):
    """Direction of every entry's update against the golden sign bits (goldgen.update_signs), per
    tensor: at least `min_agree` of the entries (all but two, or but 5 %, for tensors under 200 entries) must
    have moved the way the reference moved them.  For a network that has taken exactly one Adam
    step this is the whole update, entry by entry; the entries that may differ are those whose
    gradient lies within rounding of zero."""
    import torch
    keys = [str(k) for k in keys]
    want = np.unpackbits(np.asarray(ref_bits))
    off, bad = 0, []
    for k in keys:
        n = after[k].numel()
        got = (after[k].detach().to(torch.float64) > before[k].detach().to(torch.float64)).cpu().numpy().reshape(-1)
        w = want[off:off + n].astype(bool)
        off += n
        if k in skip or "running" in k or "num_batches" in k:
            continue
        agree = float((got == w).mean())
        need = min_agree if n >= 200 else (n - max(2, -(-n // 20))) / n       # small tensors: two entries or 5 %
        if agree < need:
            bad.append(f"{k}: {agree:.4f} of {n} entries moved the reference's way (need {need:.4f})")
    assert not bad, "\n".join(bad)
