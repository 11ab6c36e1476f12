"""Host logic of the evaluation driver (no GPU): the metric functions against the oracle's
restatement of src/evaluate.py:17-86 and src/loss.py:239-344, and date sharding -- partial
results of disjoint shards merge into exactly the unsharded metrics, files included."""
import importlib
import os

import numpy as np
import torch

from conftest import load_pkg
from oracle import streamflow_oracle as O


def _pkg():
    sf = load_pkg()
    return importlib.import_module(sf.__name__ + ".evaluation")


def test_metric_functions_match_oracle():
    ev = _pkg()
    rng = np.random.default_rng(0)
    for shape in [(40, 40), (64, 31)]:
        p = rng.normal(size=shape).astype(np.float32)
        t = (p + 0.3 * rng.normal(size=shape)).astype(np.float32)
        assert abs(ev.calculate_ssim(p, t) - O.ssim_uniform(p, t)) < 1e-7
        assert abs(ev.calculate_psnr(p, t) - O.psnr(p, t)) < 1e-6
        a, b = ev.calculate_metrics(p, t), O.calculate_metrics(p, t)
        for k in ("RMSE", "MAE", "R2", "NSE", "MAPE", "KGE"):
            assert abs(a[k] - b[k]) <= 1e-6 * max(1.0, abs(b[k])), k
    z = np.zeros((12, 12), np.float32)
    assert ev.calculate_ssim(z, z) == 1.0 and ev.calculate_psnr(z, z) == float("inf")
    assert ev.calculate_metrics(np.array([]), np.array([]))["NSE"] == 0.0
    m = ev.calculate_metrics(np.array([1.0, np.nan, 3.0]), np.array([1.0, 2.0, np.inf]))   # non-finite -> 0
    assert np.isfinite(list(m.values())).all()


class _Toy(torch.nn.Module):
    """Stand-in generator on the CPU: [B,T,C,H,W] -> [B,1,H,W] (mean over time and channels, scaled)."""
    def forward(self, x):
        return x.mean(dim=(1, 2), keepdim=False).unsqueeze(1) * 1.5 + 0.1


def _dataset(n):
    g = torch.Generator().manual_seed(3)
    out = []
    for i in range(n):
        img = torch.randn(5, 3, 24, 20, generator=g)
        ns = i % 4   # some dates without stations
        out.append({"images": img, "output_image": torch.randn(1, 24, 20, generator=g), "date": f"2020-01-{i + 1:02d}",
                    "stations": torch.randn(ns, generator=g),
                    "station_positions": torch.stack([torch.randint(0, 20, (ns,), generator=g),
                                                      torch.randint(0, 24, (ns,), generator=g)], dim=1)})
    return out


def test_sharded_evaluation_merges_to_unsharded(tmp_path):
    ev = _pkg()
    cfg = {"stations": {"kernel_size": 3}}
    e = ev.Evaluator(cfg, model=_Toy(), device="cpu")
    ds = _dataset(11)
    whole = e.evaluate_dataset(ds, output_dir=str(tmp_path / "whole"), batch_size=4)
    parts = [e.evaluate_dataset(ds, batch_size=3, shard=(r, 3))["partial"] for r in (2, 0, 1)]
    merged = ev.merge_partials(parts)
    for k in ("RMSE", "MAE", "SSIM", "PSNR", "Correlation"):
        assert abs(whole[k] - merged[k]) <= 1e-6 * max(1.0, abs(whole[k])), k   # only sums are reassociated
    for k, v in whole["station_metrics"].items():
        assert abs(v - merged["station_metrics"][k]) <= 1e-12 * max(1.0, abs(v)), k
    assert whole["station_rows"] == merged["station_rows"]
    # reference-style single-date loop gives the same numbers as batched forwarding
    single = e.evaluate_dataset(ds, batch_size=1)
    assert abs(single["RMSE"] - whole["RMSE"]) < 1e-7 and single["station_rows"] == whole["station_rows"]
    files = sorted(os.listdir(tmp_path / "whole" / "predictions"))
    assert len(files) == 22 and files[0].startswith("pred_2020-01-01")
    assert np.load(tmp_path / "whole" / "predictions" / "pred_2020-01-03.npy").shape == (24, 20)
    assert os.path.exists(tmp_path / "whole" / "stations_eval.csv") and os.path.exists(tmp_path / "whole" / "evaluation_metrics.csv")
