"""The two 262144-wide Linear layers at the benchmark's full size (input projection: 160 rows of
262144 features -> 128, src/model/transformer_module.py:73; temporal_to_spatial: 32 rows of 128 ->
262144, src/model/__init__.py:92-95), forward and all three gradients against f32 torch matmuls on
the same operands.  f32 mode: 1e-5 of the max (forward), 1e-4 (gradients; split reductions sum in
another order); bf16 mode: operands rounded to bf16 on both sides, 2e-3."""
import importlib

import pytest
import torch

from conftest import load_pkg

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def sf():
    return load_pkg()


def _rel(a, b):
    return float((a.detach().double() - b.double()).abs().max() / b.double().abs().max())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("dims", [(160, 262144, 128), (32, 128, 262144)])
def test_full_size_linear(sf, dims, dtype):
    fn = importlib.import_module(sf.__name__ + ".functional")
    M, K, N = dims
    torch.manual_seed(14)
    x = torch.randn(M, K, device=DEV)
    W = torch.randn(N, K, device=DEV) / K ** 0.5
    b = torch.randn(N, device=DEV)
    dy = torch.randn(M, N, device=DEV)
    if dtype == torch.bfloat16:
        x, W, dyr = x.bfloat16().float(), W.bfloat16().float(), dy.bfloat16().float()
    else:
        dyr = dy
    xg, Wg, bg = x.clone().requires_grad_(True), W.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = fn.LinearFn.apply(xg, Wg, bg, dtype)
    y.backward(dy)
    torch.backends.cuda.matmul.allow_tf32 = False
    yr = x.double() @ W.double().t() + b.double()
    tol_f, tol_g = (1e-5, 1e-4) if dtype == torch.float32 else (2e-3, 2e-3)
    assert _rel(y, yr) <= tol_f
    assert _rel(xg.grad, dyr.double() @ W.double()) <= tol_g
    assert _rel(Wg.grad, dyr.double().t() @ x.double()) <= tol_g
    assert _rel(bg.grad, dy.double().sum(0)) <= tol_g
