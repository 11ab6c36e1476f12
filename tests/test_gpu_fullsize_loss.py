"""Fused image + station loss at the benchmark's full size (32 samples, 400 x 400 prediction,
200 x 200 target, 64 stations per sample, 5 x 5 windows; src/loss.py:32-70,99-160,163-236):
against the CPU oracle on the same seeded inputs, and through size-independent properties --
the value does not depend on the order of the samples and the gradient moves with them, and
the gradient predicts the change of the loss along a random direction."""
import numpy as np
import pytest
import torch

from conftest import close, load_pkg
from oracle import streamflow_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
CFG = {"loss": {"metric": "MSE", "weights": {"image": 1.0, "station": 0.5}},
       "stations": {"kernel_size": 5, "metric": "MSE"}, "data": {"upscale_factor": 2}}


@pytest.fixture(scope="module")
def sf():
    return load_pkg()


def _inputs(B=32, H=400, W=400, S=64, ragged=True):
    g = torch.Generator().manual_seed(5)
    pred = torch.randn(B, 1, H, W, generator=g)
    tgt = torch.randn(B, 1, H // 2, W // 2, generator=g)
    tgt[:, :, :20, :30] = 0.0                       # exact zeros: weight 1 instead of nonzero_weight
    pos, run = [], []
    for b in range(B):
        n = S if not ragged else (0 if b == 3 else S - (b % 5))     # ragged, one empty sample
        p = torch.stack([torch.randint(0, W, (n,), generator=g), torch.randint(0, H, (n,), generator=g)], 1)
        if n >= 4:
            p[0] = torch.tensor([0, 0]); p[1] = torch.tensor([W - 1, H - 1]); p[2] = p[3]   # borders, a duplicate
        pos.append(p.to(torch.int64))
        run.append(torch.randn(n, generator=g))
    return pred, tgt, pos, run


def _hip(sf, pred, tgt, pos, run):
    p = pred.to(DEV).requires_grad_(True)
    ld = sf.CombinedLoss(CFG)(p, tgt.to(DEV), pos, run)
    ld["total"].backward()
    return ld, p.grad


def test_full_size_against_oracle(sf):
    pred, tgt, pos, run = _inputs()
    ld, grad = _hip(sf, pred, tgt, pos, run)
    pr = pred.clone().requires_grad_(True)
    lo = O.combined_loss(pr, tgt, pos, run, CFG)
    lo["total"].backward()
    for k in ("total", "image", "station"):
        close(ld[k], lo[k].detach().numpy(), 1e-5)
    close(grad, pr.grad.numpy(), 1e-5)


def test_full_size_sample_order_and_directional_derivative(sf):
    pred, tgt, pos, run = _inputs()
    ld, grad = _hip(sf, pred, tgt, pos, run)
    perm = torch.randperm(pred.shape[0], generator=torch.Generator().manual_seed(1))
    ldp, gradp = _hip(sf, pred[perm], tgt[perm], [pos[i] for i in perm], [run[i] for i in perm])
    for k in ("total", "image", "station"):
        assert abs(float(ldp[k].detach()) - float(ld[k].detach())) <= 1e-5 * abs(float(ld[k].detach()))
    close(gradp, grad[perm.to(DEV)].cpu().numpy(), 1e-5)
    # L(p + e d) - L(p - e d) = 2 e <g, d> exactly for the quadratic image term and the quadratic
    # station term (the weights depend on the target only)
    d = torch.randn(pred.shape, generator=torch.Generator().manual_seed(2))
    eps = 1e-2
    lp = float(sf.CombinedLoss(CFG)((pred + eps * d).to(DEV), tgt.to(DEV), pos, run)["total"])
    lm = float(sf.CombinedLoss(CFG)((pred - eps * d).to(DEV), tgt.to(DEV), pos, run)["total"])
    lin = 2 * eps * float((grad.double().cpu() * d.double()).sum())
    assert abs((lp - lm) - lin) <= 2e-3 * abs(lin) + 1e-6


def test_split_forward_backward_against_the_fused_call(sf):
    """sf_image_station_loss_fwd / _bwd (one launch each, what the autograd Function uses) against
    sf_image_station_loss (both back to back, upstream 1): same losses bit for bit; the gradient
    with an upstream factor is the fused gradient times that factor; with and without the saved
    station differences; two runs are bit-identical (no atomics anywhere); the prediction check
    sets its flag only for an Inf / NaN."""
    import importlib
    SF = importlib.import_module(sf.__name__ + ".functional")
    ops = sf.ops
    pred, tgt, pos, run = _inputs()
    p, t = pred.to(DEV), tgt.to(DEV)
    dpos, drun, doffs, S = SF.pack_stations(pos, run, p.device)
    args = (p, t, dpos, drun, doffs, S, 0, 10.0, 1e-6, 5, 0, 1.0, 0.5)
    ref_l, ref_g = ops.image_station_loss(*args)
    assert float(ref_l[3]) == 1.0
    flag = torch.zeros(1, device=DEV, dtype=torch.int32)
    losses, sdiff = ops.image_station_loss_fwd(*args, nan_flag=flag)
    assert torch.equal(losses[:3], ref_l[:3]) and int(flag) == 0
    losses2, _ = ops.image_station_loss_fwd(*args, nan_flag=flag)
    assert torch.equal(losses2, losses)
    up = torch.tensor([2.5], device=DEV)
    g1 = ops.image_station_loss_bwd(*args, losses, sdiff=sdiff, upstream=up)
    g2 = ops.image_station_loss_bwd(*args, losses, sdiff=None, upstream=up)
    g3 = ops.image_station_loss_bwd(*args, losses, sdiff=sdiff, upstream=None)
    assert torch.equal(g1, ops.image_station_loss_bwd(*args, losses, sdiff=sdiff, upstream=up))
    close(g1, (2.5 * ref_g).cpu().numpy(), 1e-6)
    close(g2, g1.cpu().numpy(), 1e-6)
    assert torch.equal(g3, ref_g)
    # a single NaN / Inf anywhere in pred raises the flag (and only then)
    for bad in (float("nan"), float("-inf")):
        q = p.clone()
        q[17, 0, 399, 398] = bad
        flag.zero_()
        ops.image_station_loss_fwd(q, *args[1:], nan_flag=flag)
        assert int(flag) == 1
