"""Temporal modules at the benchmark's width and batch (32 sequences of 5 steps, hidden 128, four
heads, feed-forward 512; src/model/transformer_module.py:36-120, src/model/lstm_module.py:20-77):
forward and every gradient against autograd through the CPU oracle with the module's own
parameters.  f32: 1e-5 of the max (forward), 1e-4 (gradients)."""
import pytest
import torch

from conftest import close, load_pkg
from oracle import streamflow_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def sf():
    return load_pkg()


@pytest.mark.parametrize("kind", ["tr", "lstm"])
def test_full_size_temporal_against_oracle(sf, kind):
    B, T, D_in, Hd = 32, 5, 512, 128
    torch.manual_seed(15)
    if kind == "tr":
        mod = sf.TransformerModule(D_in, Hd, num_layers=2, num_heads=4, dropout=0.0).to(DEV)
    else:
        mod = sf.LSTMModule(D_in, Hd, num_layers=2, dropout=0.0).to(DEV)
    with torch.no_grad():                       # biases / norms away from their trivial initial values
        for n, p in mod.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    x0 = torch.randn(B, T, D_in)
    gy = torch.randn(B, Hd)
    mod.train()
    x = x0.to(DEV).requires_grad_(True)
    y = mod(x)
    (y * gy.to(DEV)).sum().backward()
    sd = {"m." + k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point and k.split(".")[-1] != "pe")
          for k, v in mod.state_dict().items()}
    xr = x0.clone().requires_grad_(True)
    yr = O.transformer_forward(xr, sd, "m.", 2, 4) if kind == "tr" else O.lstm_forward(xr, sd, "m.", 2)
    (yr * gy).sum().backward()
    close(y, yr.detach().numpy(), 1e-5)
    close(x.grad, xr.grad.numpy(), 1e-4)
    for n, p in mod.named_parameters():
        close(p.grad, sd["m." + n].grad.numpy(), 1e-4)
