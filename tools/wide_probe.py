"""A few launches of one 3x3 conv shape on the LDS-DMA kernel (staged weights), for counter passes:
    rocprofv3 --pmc ... -- python tools/wide_probe.py [N H C K [dgrad]]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("tcn-transformer-gan-hybrid-model-for-predicting-streamflow-in-ungauged-basins_amd")
ops = pkg.ops
torch.manual_seed(0)
N, H, C, K = [int(v) for v in sys.argv[1:5]] if len(sys.argv) >= 5 else (160, 100, 128, 256)
x = torch.randn(N, H, H, C, device="cuda").bfloat16()
wp = (torch.randn(K, 9, C, device="cuda") * 0.05).bfloat16()
ws = ops.conv3x3_stage_weights(wp, C, K)
out = torch.empty(N, H, H, K, device="cuda", dtype=torch.bfloat16)
for _ in range(6):
    ops.conv3x3(x, ws, out, K)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.conv3x3(x, ws, out, K)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"C {C} K {K} N {N} HW {H} ms {ms:.4f} TF/s {2 * N * H * H * C * K * 9 / ms / 1e9:.0f}", flush=True)
