"""Times the last encoder layer's pooling forward (with and without bin moments) and the two
BatchNorm-backward reductions at the bench shape (160 x 100 x 100 x 256 bf16 -> 32 x 32)."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sf = importlib.import_module("tcn-transformer-gan-hybrid-model-for-predicting-streamflow-in-ungauged-basins_amd")
ops = sf.ops
DEV = "cuda:0"


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    N, H, W, C, OH = 160, 100, 100, 256, 32
    x = torch.randn(N, H, W, C, device=DEV).to(torch.bfloat16)
    dy = torch.randn(N, OH, OH, C, device=DEV).to(torch.bfloat16)
    gamma, beta = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV) * 0.3
    ws = ops.bn_ws(C, DEV)
    ops.bn_stats(x, N * H * W, C, ws)
    st = ops.bn_finalize(ws[:2 * C], N * H * W, C, C, gamma, beta, torch.zeros(C, device=DEV), torch.ones(C, device=DEV), True)
    y, pos, xh = ops.bn_relu_avgpool_fwd_moments(x, st, C, OH, OH, torch.bfloat16)
    print("stats           %.0f us" % timeit(lambda: ops.bn_stats(x, N * H * W, C, ws)))
    print("fwd             %.0f us" % timeit(lambda: ops.bn_relu_avgpool_fwd(x, st, C, OH, OH, torch.bfloat16)))
    print("fwd + moments   %.0f us" % timeit(lambda: ops.bn_relu_avgpool_fwd_moments(x, st, C, OH, OH, torch.bfloat16)))
    print("bwd from x      %.0f us" % timeit(lambda: ops.bn_backward(1, x, dy, st, gamma, C, True, OH, OH)))
    print("bwd from bins   %.0f us" % timeit(lambda: ops.bn_backward(1, x, dy, st, gamma, C, True, OH, OH, moments=(pos, xh))))


if __name__ == "__main__":
    main()
