"""Fused Adam (src/train.py:56,66-76: torch.optim.Adam, lr 1e-4 / 2e-5, betas (0.9, 0.999) /
(0.5, 0.999), eps 1e-8, no weight decay) on a flat buffer the size of the benchmark's generator
(71 M parameters): three steps against torch.optim.Adam on the same device, including the
gradient scale that folds the data-parallel 1 / world factor into the launch, and the
NaN / Inf flag over the same buffer.  Parameters 1e-6 of their max (f32 arithmetic in another
order), moments 1e-6."""
import pytest
import torch

from conftest import load_pkg

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def sf():
    return load_pkg()


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


@pytest.mark.parametrize("hyper", [(1e-4, 0.9, 0.999, 1.0), (2e-5, 0.5, 0.999, 0.125)])
def test_full_size_adam_against_torch(sf, hyper):
    ops = sf.ops
    lr, b1, b2, gs = hyper
    n = 71_000_003                      # not a multiple of any vector width
    torch.manual_seed(19)
    p = torch.randn(n, device=DEV)
    ref = torch.nn.Parameter(p.clone())
    opt = torch.optim.Adam([ref], lr=lr, betas=(b1, b2), eps=1e-8, foreach=False, fused=False)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    flag = torch.zeros(1, device=DEV, dtype=torch.int32)
    for step in (1, 2, 3):
        g = torch.randn(n, device=DEV) * (10.0 ** (step - 2))
        ops.nan_inf_flag(g, flag)
        ops.adam(p, g, m, v, lr, b1, b2, 1e-8, step, grad_scale=gs)
        ref.grad = g * gs
        opt.step()
    st = opt.state[ref]
    assert int(flag.item()) == 0
    assert _rel(p, ref.detach()) <= 1e-6
    assert _rel(m, st["exp_avg"]) <= 1e-6 and _rel(v, st["exp_avg_sq"]) <= 1e-6
    g[n - 1] = float("inf")
    ops.nan_inf_flag(g, flag)
    assert int(flag.item()) != 0
