"""The encoder's second conv (and its data gradient) on the wide LDS-DMA kernel: plain packed
weights against the staged (contiguous copy pieces) form; the outputs must be identical."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("tcn-transformer-gan-hybrid-model-for-predicting-streamflow-in-ungauged-basins_amd")
ops = pkg.ops
torch.manual_seed(0)


def run(N, H, C, K, n=10):
    x = torch.randn(N, H, H, C, device="cuda").bfloat16()
    wp = (torch.randn(K, 9, C, device="cuda") * 0.05).bfloat16()
    ws = ops.conv3x3_stage_weights(wp, C, K)
    outs = []
    for name, w in (("plain", wp), ("staged", ws)):
        out = torch.empty(N, H, H, K, device="cuda", dtype=torch.bfloat16)
        for _ in range(3):
            ops.conv3x3(x, w, out, K)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            ops.conv3x3(x, w, out, K)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        print(f"C {C} K {K} N {N} HW {H} {name:7s} ms {ms:.3f} TF/s {2 * N * H * H * C * K * 9 / ms / 1e9:.0f}", flush=True)
        outs.append(out)
    print("identical:", torch.equal(outs[0], outs[1]), flush=True)


run(160, 100, 128, 256)
run(160, 100, 256, 128)
run(160, 200, 16, 128, n=5)
run(3, 37, 48, 136)
