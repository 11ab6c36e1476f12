"""The two large configurations at their stated extent: ten years of daily frames (3655 dates of
10 x 200 x 200, 5.8 GB resident in HBM, 64 stations) and batch inference over all 3650 window
dates at 400 x 400 output (src/dataset.py:95-114, src/evaluate.py:149-335).

  * the resident dataset builds and serves all 3650 windows; a shuffled epoch split over 8 ranks
    covers every window exactly once;
  * the evaluation of all 3650 dates in two date shards, merged, equals the unsharded evaluation
    (sharding by date needs no exchange: the partial results are order-free);
  * 64 dates spread over the ten years equal the CPU oracle's batch-1 evaluation of the same
    samples (1e-4, the tolerance of the 8-date test).

The frames are a short synthetic series (blobs + filters, as the benchmark's generator) repeated
with a slowly varying gain, which costs seconds instead of the minutes the generator would need
for ten years; the statistics, windows and station tables still span all 3655 dates."""
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
DAYS, BASE = 3655, 61


def _ten_years(synth):
    import datetime
    import pandas as pd
    base = synth.make_days(BASE, 10, 200, 200, seed=4)
    gain = (1.0 + 0.25 * np.sin(np.arange(DAYS) * 0.0172)).astype(np.float32)       # a yearly cycle
    frames = [base[d % BASE] * gain[d] for d in range(DAYS)]
    rs = np.random.RandomState(9)
    lon, lat = rs.uniform(100, 110, 64), rs.uniform(25, 35, 64)
    start = datetime.date(2015, 1, 1)
    dates = [(start + datetime.timedelta(days=d)).strftime("%Y%m%d") for d in range(DAYS)]
    ts = [int(d) for d in dates]
    stations = {f"station_{s:03d}": pd.DataFrame({"timestamp": ts, "lon": lon[s], "lat": lat[s],
                                                  "runoff": np.abs(rs.standard_normal(DAYS)) * 40.0 * gain})
                for s in range(64)}
    return frames, dates, stations


def test_ten_years_resident_and_all_dates_evaluated_in_shards():
    sf = load_pkg()
    synth = importlib.import_module(sf.__name__ + ".synth")
    evaluation = importlib.import_module(sf.__name__ + ".evaluation")
    frames, dates, stations = _ten_years(synth)
    ds = sf.ResidentStreamflowDataset.from_arrays(frames, dates, stations, CFG)
    del frames
    n = len(ds)
    assert n == DAYS - 5 == 3650 and tuple(ds.frames.shape) == (DAYS, 10, 200, 200)
    # an epoch over 8 ranks: disjoint slices of one permutation, complete up to the tail that does not divide
    dataset = importlib.import_module(sf.__name__ + ".dataset")
    shards = [np.concatenate(dataset.window_shards(n, 32, True, 0, 3, r, 8, False)) for r in range(8)]
    allidx = np.concatenate(shards)
    assert len(set(allidx.tolist())) == len(allidx) == n - n % 8 and len({len(s_) for s_ in shards}) == 1
    b = next(iter(ds.loader(32, shuffle=True, seed=0, epoch=3, rank=5, world=8)))
    assert tuple(b["images"].shape) == (32, 5, 10, 200, 200) and tuple(b["output_image"].shape) == (32, 1, 200, 200)
    first = int(shards[5][0])
    assert torch.equal(b["images"][0], ds[first]["images"]) and torch.equal(b["output_image"][0], ds[first]["output_image"])
    m = sf.StreamflowPredictionModel(CFG).to("cuda")
    sd = load_gold(m, "fs_eval_g")
    ev = sf.Evaluator(CFG, model=m)
    half = n // 2
    pa = ev.evaluate_partial_resident(ds, range(0, half), batch_size=32)
    pb = ev.evaluate_partial_resident(ds, range(half, n), batch_size=32)
    whole = evaluation.merge_partials([ev.evaluate_partial_resident(ds, range(n), batch_size=32)])
    merged = evaluation.merge_partials([pb, pa])                       # shard order must not matter
    keys, skeys = ("RMSE", "MAE", "SSIM", "PSNR", "Correlation"), ("RMSE", "MAE", "R2", "NSE", "MAPE", "KGE")
    assert pa["npx"] + pb["npx"] == n * 400 * 400 and len(merged["station_rows"]) == n * 64
    np.testing.assert_allclose([merged[k] for k in keys], [whole[k] for k in keys], rtol=1e-7)
    np.testing.assert_allclose([merged["station_metrics"][k] for k in skeys], [whole["station_metrics"][k] for k in skeys],
                               rtol=1e-7)
    assert np.isfinite([whole[k] for k in keys]).all()
    # 64 dates spread over the ten years against the oracle's batch-1 evaluation
    sub = [int(i) for i in np.linspace(0, n - 1, 64).round()]
    got = evaluation.merge_partials([ev.evaluate_partial_resident(ds, sub, batch_size=16)])
    samples = []
    for i in sub:
        s = ds[i]
        samples.append({"images": s["images"].cpu(), "output_image": s["output_image"].cpu(),
                        "stations": s["stations"], "station_positions": s["station_positions"]})
    want = O.evaluate_samples(samples, {k: v.cpu() for k, v in sd.items()}, CFG)
    np.testing.assert_allclose([got[k] for k in keys], [want[k] for k in keys], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose([got["station_metrics"][k] for k in skeys], [want["station_metrics"][k] for k in skeys],
                               rtol=1e-4, atol=1e-6)
