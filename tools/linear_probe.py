"""Times of the two skinny Linear layers (262144 <-> 128) per GEMM, bf16, B = 32, T = 5."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sf = importlib.import_module("tcn-transformer-gan-hybrid-model-for-predicting-streamflow-in-ungauged-basins_amd")
ops = sf.ops
bf = torch.bfloat16


def timeit(name, fn, nbytes, n=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / n
    print(f"{name:34s} {t * 1e3:7.1f} us   {nbytes / t / 1e6:6.0f} GB/s of {nbytes / 1e6:.0f} MB", flush=True)


K = 262144
import importlib as _il
SF = _il.import_module(sf.__name__ + ".functional")
for tag, M, Nf, Kd in (("input_proj (160 x 262144 -> 128)", 160, 128, K), ("temporal_to_spatial (32 x 128 -> 262144)", 32, K, 128)):
    print(tag)
    x = torch.randn(M, Kd, device="cuda").to(bf)
    W = torch.randn(Nf, Kd, device="cuda").to(bf)
    dy = torch.randn(M, Nf, device="cuda").to(bf)
    bk = 32
    ns = max(1, min(256, Kd // (bk * 16)))
    timeit("  forward", lambda: SF._split_gemm(W, x, Nf, M, Kd, Kd, Kd, Nf, False, ns, None), 2 * (x.numel() + W.numel()) + 4 * M * Nf)
    ns2 = max(1, min(256, Nf // (bk * 16)))
    timeit("  dx", lambda: SF._split_gemm(W, dy, Kd, M, Nf, Kd, Nf, Kd, True, ns2, None), 2 * (dy.numel() + W.numel()) + 4 * M * Kd)
    dW = torch.empty(Nf, Kd, device="cuda")
    timeit("  dW", lambda: ops.linear_gemm(dy, x, dW, Nf, Kd, M, Nf, Kd, Kd, True, True, ops.OUT_IJ, 1), 2 * (dy.numel() + x.numel()) + 4 * dW.numel())
    timeit("  cast W f32 -> bf16", lambda: ops.cast(dW, bf), 6 * dW.numel())
