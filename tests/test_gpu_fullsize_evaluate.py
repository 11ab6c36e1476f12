"""Inference / evaluation at the benchmark's inference configuration (10 channels, 200 x 200 ->
400 x 400, Transformer, hidden 128, 64 stations, 5 x 5 windows; src/evaluate.py:149-335) over a
device-resident dataset of 13 synthetic dates (8 evaluation dates): frames encoded once per date,
image metrics reduced on the device, against the CPU oracle's per-date batch-1 evaluation of the
same normalised samples with the same weights.  RMSE / MAE / SSIM / PSNR / correlation and the
station metrics 1e-4 relative (f32 forward within 1e-5; float64 SSIM window sums here, float32
scipy filters in the oracle)."""
import importlib

import numpy as np
import pytest
import torch

from conftest import load_pkg
from oracle import streamflow_oracle as O
from test_gpu_parity import load_gold

pytestmark = pytest.mark.gpu
CFG = {"data": {"upscale_factor": 2, "image_size": [200, 200], "window_size": 5, "region": [100, 110, 25, 35]},
       "model": {"input_channels": 10, "hidden_dim": 128, "num_layers": 2, "temporal_module": "Transformer"},
       "stations": {"kernel_size": 5}}


def test_full_size_resident_evaluation_against_oracle():
    sf = load_pkg()
    synth = importlib.import_module(sf.__name__ + ".synth")
    raw, dates, st = synth.make_dataset_inputs(13, seed=3)
    ds = sf.ResidentStreamflowDataset.from_arrays(list(raw), dates, st, CFG)
    n = len(ds)
    assert n == 8
    m = sf.StreamflowPredictionModel(CFG).to("cuda")
    sd = load_gold(m, "fs_eval_g")
    ev = sf.Evaluator(CFG, model=m)
    samples = []
    for i in range(n):
        s = ds[i]
        samples.append({"images": s["images"].cpu(), "output_image": s["output_image"].cpu(),
                        "stations": s["stations"], "station_positions": s["station_positions"]})
    want = O.evaluate_samples(samples, {k: v.cpu() for k, v in sd.items()}, CFG)
    keys, skeys = ("RMSE", "MAE", "SSIM", "PSNR", "Correlation"), ("RMSE", "MAE", "R2", "NSE", "MAPE", "KGE")
    for bs in (3, 8):
        res = ev.evaluate_dataset(ds, batch_size=bs)
        np.testing.assert_allclose([res[k] for k in keys], [want[k] for k in keys], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose([res["station_metrics"][k] for k in skeys],
                                   [want["station_metrics"][k] for k in skeys], rtol=1e-4, atol=1e-6)
