"""Time of the fused image + station loss (forward and gradient) at the bench shape and at a
batch large enough not to be launch bound; algorithmic bytes = pred read + target read + dpred write."""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sf = importlib.import_module("tcn-transformer-gan-hybrid-model-for-predicting-streamflow-in-ungauged-basins_amd")
ops = sf.ops
SF = importlib.import_module(sf.__name__ + ".functional")
for B in (32, 256):
    pred = torch.randn(B, 1, 400, 400, device="cuda")
    tgt = torch.randn(B, 1, 200, 200, device="cuda")
    rs = np.random.RandomState(0)
    S = 64
    packed = SF.PackedStations.from_host(rs.randint(0, 200, size=(B * S, 2)), rs.standard_normal(B * S).astype(np.float32),
                                         np.full(B, S), "cuda")

    def fn():
        return ops.image_station_loss(pred, tgt, packed.pos, packed.run, packed.offs, packed.S, 0, 10.0, 1e-6, 5, 0, 1.0, 0.5)

    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        fn()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 50
    nbytes = pred.numel() * 8 + tgt.numel() * 4 + B * S * 12
    print(f"B={B}: {t * 1e3:.1f} us per call (2 launches), {nbytes / 1e6:.1f} MB algorithmic -> {nbytes / t / 1e6:.0f} GB/s", flush=True)
