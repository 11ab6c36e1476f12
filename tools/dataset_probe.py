"""Bandwidth of the three resident-dataset kernels at the C4 frame shape (10 x 200 x 200)."""
import importlib
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sf = importlib.import_module("tcn-transformer-gan-hybrid-model-for-predicting-streamflow-in-ungauged-basins_amd")
ops = sf.ops
D, C, H, W, B, T = 1024, 10, 200, 200, 32, 5
frames = torch.randn(D, C, H, W, device="cuda")
a = torch.zeros(C, device="cuda", dtype=torch.float64)
b = torch.ones(C, device="cuda", dtype=torch.float64)
rows = torch.from_numpy(np.random.RandomState(0).randint(0, D, size=B * T)).cuda()
out = torch.empty(B * T, C * H * W, device="cuda")


def timeit(fn, n=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


fb = frames.numel() * 4
t = timeit(lambda: ops.dataset_plane_stats(frames, None, None))
print(f"plane_stats  {t:.3f} ms  {fb / t / 1e6:.0f} GB/s (read {fb / 1e9:.2f} GB)")
t = timeit(lambda: ops.dataset_transform(frames, None, None, None, a, b, 0.1))
print(f"transform    {t:.3f} ms  {2 * fb / t / 1e6:.0f} GB/s (read+write)")
gb = out.numel() * 4
t = timeit(lambda: ops.gather_frames(frames, C * H * W, D, rows, C * H * W, out), 50)
print(f"gather B=32  {t:.3f} ms  {2 * gb / t / 1e6:.0f} GB/s (read+write {2 * gb / 1e6:.0f} MB)")
