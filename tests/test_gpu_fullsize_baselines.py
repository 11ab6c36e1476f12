"""Baseline comparison models (src/model/baseline_models.py:21-214,340-352) at the benchmark's input
size (32 windows of 5 x 10 x 200 x 200 -> 400 x 400, hidden 128) in f32 mode: training-mode
forward against the CPU oracle with the module's own parameters (1e-5 of the max), BatchNorm
running buffers (1e-5), and the input gradient through autograd of the oracle
(conftest.close_robust: ReLU decisions within rounding of zero)."""
import importlib

import pytest
import torch

from conftest import close, close_robust, load_pkg
from oracle import streamflow_oracle as O

pytestmark = pytest.mark.gpu
CFG = {"model": {"input_channels": 10, "hidden_dim": 128, "num_layers": 2},
       "data": {"image_size": [200, 200], "upscale_factor": 2, "window_size": 5}}


@pytest.fixture(scope="module")
def base():
    sf = load_pkg()
    return importlib.import_module(sf.__name__ + ".baselines")


@pytest.mark.parametrize("name", ["LSTM", "BP", "Attention-GRU"])
def test_full_size_baseline_against_oracle(base, name):
    torch.manual_seed(21)
    m = base.build_baseline_model(name, CFG).to("cuda")
    if name == "LSTM":
        m.lstm.dropout = 0.0
    elif name == "Attention-GRU":
        m.gru.dropout = 0.0
    else:
        m.mlp[2].p = m.mlp[5].p = 0.0
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    x0 = torch.randn(32, 5, 10, 200, 200)
    u = torch.randn(32, 1, 400, 400)
    m.train()
    x = x0.cuda().requires_grad_(True)
    y = m(x)
    (y * u.cuda()).sum().backward()
    xr = x0.clone().requires_grad_(True)
    want, new = O.baseline_forward(xr, sd, name, 400, 400, training=True)
    (want * u).sum().backward()
    close(y, want.detach().numpy(), 1e-5)
    now = m.state_dict()
    for k, v in new.items():
        close(now[k].float(), v.detach().float().numpy(), 1e-5)
    # 64 M gradient elements: at most one in 10^4 beyond 1e-3, and a flipped decision right at an
    # extreme element may move it by up to a fifth of the tensor's max; relative L2 error 1e-3
    close_robust(x.grad, xr.grad.numpy(), 1e-3, frac=1e-4, loose=0.2)
    d = x.grad.detach().cpu().double() - xr.grad.double()
    assert float(d.norm() / xr.grad.double().norm()) <= 1e-3
