"""Baseline models (src/model/baseline_models.py: LightweightCNNReducer, PureLSTMModel, PureBPModel)
on the HIP kernels against the reference's outputs and gradients (tests/golden/baselines.npz) and
the CPU restatement.  f32: forward <= 1e-5 of the max, gradients through ReLU decisions with the
robust comparison; bf16 at the stated 3e-2."""
import importlib

import numpy as np
import pytest
import torch

import goldgen as G
from conftest import close, close_robust, golden, load_pkg
from oracle import streamflow_oracle as O
from test_gpu_parity import load_gold

pytestmark = pytest.mark.gpu
CFG = {"model": {"input_channels": 3, "hidden_dim": 32, "num_layers": 2},
       "data": {"image_size": [64, 64], "upscale_factor": 2, "window_size": 5}}


@pytest.fixture(scope="module")
def base():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    sf = load_pkg()
    return importlib.import_module(sf.__name__ + ".baselines")


def _model(base, name):
    m = base.build_baseline_model(name, CFG).to("cuda")
    assert [k for k in m.state_dict()] == [str(k) for k in golden("baselines")[f"{name}_keys"]]
    load_gold(m, "base_" + name)
    if name == "LSTM":
        m.lstm.dropout = 0.0
    elif name == "Attention-GRU":
        m.gru.dropout = 0.0
    else:
        m.mlp[2].p = m.mlp[5].p = 0.0
    return m


@pytest.mark.parametrize("name", ["LSTM", "BP", "Attention-GRU"])
def test_baseline_forward_backward_f32(base, name):
    g = golden("baselines")
    m = _model(base, name)
    x = G.gold_tensor("base/x", (2, 5, 3, 64, 64)).cuda()
    u = G.gold_tensor("base/u", (2, 1, 128, 128)).cuda()
    m.eval()
    with torch.no_grad():
        close(m(x), g[f"{name}_eval"], 1e-5)
    m.train()
    y = m(x)
    close(y, g[f"{name}_train"], 1e-5)
    (y * u).sum().backward()
    sd = m.state_dict()
    close(sd["reducer.encoder.7.running_mean"], g[f"{name}_rm7"], 1e-5)
    close(sd["reducer.encoder.1.running_var"], g[f"{name}_rv1"], 1e-5)
    assert int(sd["reducer.encoder.1.num_batches_tracked"]) == 1
    grads = {k: p.grad for k, p in m.named_parameters()}
    for k in ("reducer.encoder.0.weight", "reducer.encoder.6.weight", "reducer.encoder.4.weight", "decoder.0.bias",
              "decoder.2.weight", "head_proj.bias"):
        close_robust(grads[k], g[f"{name}_g_{k}"], 1e-3)
    keys = [str(k) for k in g[f"{name}_gkeys"]]
    cs = G.checksum({k: v.cpu() for k, v in grads.items()}, keys)
    got, want = np.stack([cs[k] for k in keys]), g[f"{name}_gsum"]
    # abs-sums of every gradient; the attention key bias has a mathematically zero gradient (a
    # constant added to every score of a row leaves the softmax unchanged): rounding noise only
    np.testing.assert_allclose(got[:, 1], want[:, 1], rtol=2e-3, atol=1e-7 * want[:, 1].max())


@pytest.mark.parametrize("name", ["LSTM", "BP", "Attention-GRU"])
def test_baseline_bf16_and_oracle(base, name):
    g = golden("baselines")
    m = _model(base, name)
    x = G.gold_tensor("base/x", (2, 5, 3, 64, 64))
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    want, _ = O.baseline_forward(x, sd, name, 128, 128, training=False)
    m.eval()
    with torch.no_grad():
        close(m(x.cuda()), want.numpy(), 1e-5)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y16 = m(x.cuda())
    close(y16, g[f"{name}_eval"], 3e-2)
    m.train()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = m(x.cuda())
    y.float().sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
    close(y, g[f"{name}_train"], 3e-2)


def test_gru_dropout_reproducible(base):
    """Inter-layer dropout of the GRU: same seed, same mask (outputs equal up to the atomics order
    of the split-K Linear layers); gradients finite; p = 0 differs."""
    torch.manual_seed(0)
    m = base.AttentionGRUModel(CFG).cuda().train()
    x = torch.randn(2, 5, 3, 64, 64, device="cuda")
    torch.manual_seed(7)
    a = m(x)
    torch.manual_seed(7)
    b = m(x)
    assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max())
    a.sum().backward()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters())
    m.gru.dropout = 0.0
    assert float((m(x) - a).abs().max()) > 1e-3 * float(a.abs().max())


def test_relu_dropout_and_factory(base):
    torch.manual_seed(0)
    x = torch.randn(64, 512, device="cuda", requires_grad=True)
    y = base.ReluDropoutFn.apply(x, 0.2, 1234)
    frac = float(((y == 0) & (x > 0)).sum()) / float((x > 0).sum())
    assert 0.17 < frac < 0.23                                               # about p of the positive inputs dropped
    assert torch.allclose(y[(y != 0)], (x / 0.8)[(y != 0)])
    assert torch.equal(base.ReluDropoutFn.apply(x, 0.2, 1234), y)           # same seed, same mask
    y.sum().backward()
    assert torch.allclose(x.grad, torch.where(y > 0, torch.full_like(y, 1.25), torch.zeros_like(y)))
    assert torch.equal(base.ReluDropoutFn.apply(x, 0.0, 0), torch.relu(x))
    with pytest.raises(ValueError):
        base.build_baseline_model("nope", CFG)
