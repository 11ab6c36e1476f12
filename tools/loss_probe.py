"""Time of the image + station loss at the bench shape and at a batch large enough not to be launch
bound: the forward launch (reads pred and target), the backward launch (reads both, writes the
gradient), each bracketed by HIP events over back-to-back calls.  Algorithmic bytes: forward = pred +
target + 12 B per station; backward = pred + target + gradient + 12 B per station.  Writes a JSON
summary when a path is given:   python tools/loss_probe.py [out.json]"""
import importlib
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sf = importlib.import_module("tcn-transformer-gan-hybrid-model-for-predicting-streamflow-in-ungauged-basins_amd")
ops = sf.ops
SF = importlib.import_module(sf.__name__ + ".functional")
HBM_PEAK_GBS = 8000.0
rows = {}


def timed(fn, n=200):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3     # us


for B in (32, 256):
    pred = torch.randn(B, 1, 400, 400, device="cuda")
    tgt = torch.randn(B, 1, 200, 200, device="cuda")
    rs = np.random.RandomState(0)
    S = 64
    packed = SF.PackedStations.from_host(rs.randint(0, 200, size=(B * S, 2)), rs.standard_normal(B * S).astype(np.float32),
                                         np.full(B, S), "cuda")
    args = (pred, tgt, packed.pos, packed.run, packed.offs, packed.S, 0, 10.0, 1e-6, 5, 0, 1.0, 0.5)
    flag = torch.zeros(1, device="cuda", dtype=torch.int32)
    losses, sdiff = ops.image_station_loss_fwd(*args, nan_flag=flag)
    up = torch.ones(1, device="cuda")
    t_f = timed(lambda: ops.image_station_loss_fwd(*args, nan_flag=flag))
    t_b = timed(lambda: ops.image_station_loss_bwd(*args, losses, sdiff=sdiff, upstream=up))
    b_f = pred.numel() * 4 + tgt.numel() * 4 + B * S * 12
    b_b = pred.numel() * 8 + tgt.numel() * 4 + B * S * 12
    row = {"forward_us": round(t_f, 2), "backward_us": round(t_b, 2), "forward_bytes": b_f, "backward_bytes": b_b,
           "forward_GBs": round(b_f / t_f / 1e3, 1), "backward_GBs": round(b_b / t_b / 1e3, 1),
           "both_GBs": round((b_f + b_b) / (t_f + t_b) / 1e3, 1),
           "frac_of_hbm_peak": round((b_f + b_b) / (t_f + t_b) / 1e3 / HBM_PEAK_GBS, 4)}
    rows[f"B={B}"] = row
    print(f"B={B}: forward {t_f:.1f} us ({row['forward_GBs']:.0f} GB/s), backward {t_b:.1f} us ({row['backward_GBs']:.0f} GB/s), "
          f"both {row['both_GBs']:.0f} GB/s = {100 * row['frac_of_hbm_peak']:.1f} % of {HBM_PEAK_GBS:.0f} GB/s", flush=True)
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as f:
        json.dump({"what": "image + station loss, 400x400 predictions, 200x200 targets, 64 stations per sample, k = 5; one launch "
                           "forward (with the Inf/NaN check of the predictions), one launch backward; event-timed means of 200 "
                           "back-to-back calls (includes launch gaps)", "peak_GBs": HBM_PEAK_GBS, "shapes": rows}, f, indent=1)
