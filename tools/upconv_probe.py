"""Time of the decoder's resize-free conv1 passes at the benchmark shape (32 source maps of 32 x 32 with 9 x 64 tap
values per pixel -> 400 x 400 x 64): sf_upconv3_fwd and sf_upconv3_bwd."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("tcn-transformer-gan-hybrid-model-for-predicting-streamflow-in-ungauged-basins_amd")
ops = pkg.ops
B, h, K, OH = 32, 32, 64, 400
y9 = torch.randn(B, h, h, 9, K, device="cuda")
bias = torch.randn(K, device="cuda")
dz = torch.randn(B, OH, OH, K, device="cuda").bfloat16()


def t(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print("upconv3_fwd %.1f us   upconv3_bwd %.1f us" % (t(lambda: ops.upconv3_fwd(y9, bias, OH, OH, torch.bfloat16, relu=True)),
                                                     t(lambda: ops.upconv3_bwd(dz, h, h, torch.bfloat16))), flush=True)
