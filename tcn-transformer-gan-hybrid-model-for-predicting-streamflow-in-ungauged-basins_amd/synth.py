"""Synthetic benchmark inputs with the distribution of the reference's dummy-data generator
(generate_dummy_data.py:129-254) followed by the dataset transform (src/dataset.py:433,472,
490-491,503): moving Gaussian blobs smoothed with gaussian_filter(sigma=2), 10 correlated
channels, clip [0, 500], |ch0| < 0.1 -> 0, per-channel z-score; stations uniform in the region,
runoff = 5x5 mean of channel 0 + N(0, 10^2) clipped at 0, z-scored.  Everything is generated in
memory (no files) and windows of T consecutive days predict the next day."""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import torch


def _blob_field(h, w, centers, amps, sigmas):
    yy, xx = np.mgrid[0:h, 0:w]
    f = np.zeros((h, w), dtype=np.float64)
    for (cy, cx), a, s in zip(centers, amps, sigmas):
        f += a * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * s ** 2))
    return f


def make_days(days: int, channels: int, height: int, width: int, seed: int = 0) -> np.ndarray:
    from scipy.ndimage import gaussian_filter
    rs = np.random.RandomState(42)  # blob behaviour is seeded with 42 in the reference
    blobs = []
    for _ in range(4):
        y = rs.uniform(height * 0.1, height * 0.9)
        x = rs.uniform(width * 0.1, width * 0.9)
        speed = rs.uniform(0.5, 3)
        direction = rs.uniform(0, 2 * np.pi)
        blobs.append(dict(x=x, y=y, vx=speed * np.cos(direction), vy=speed * np.sin(direction),
                          sigma=rs.uniform(5, 20), amp=rs.uniform(50, 150)))
    rs = np.random.RandomState(seed + 1)
    out = np.zeros((days, channels, height, width), dtype=np.float32)
    for d in range(days):
        for b in blobs:
            b["x"] = (b["x"] + b["vx"]) % width
            b["y"] = (b["y"] + b["vy"]) % height
        base = _blob_field(height, width, [(b["y"], b["x"]) for b in blobs], [b["amp"] for b in blobs],
                           [b["sigma"] for b in blobs])
        base = gaussian_filter(base, sigma=2.0)
        g5 = gaussian_filter(base, sigma=5.0)
        for ch in range(channels):
            if ch == 0:
                out[d, ch] = base
            elif ch == 1:
                out[d, ch] = gaussian_filter(base, sigma=3.0)
            else:
                wgt = 0.5 + 0.5 * np.sin(2 * np.pi * ch / channels)
                out[d, ch] = wgt * base + (1 - wgt) * g5 + rs.randn(height, width) * 5
    return np.clip(out, 0, 500)


def make_batches(n_batches: int, batch: int, channels: int = 10, height: int = 200, width: int = 200,
                 window: int = 5, stations: int = 64, region=(100.0, 110.0, 25.0, 35.0), seed: int = 0,
                 lonlat_to_pixel=None) -> List[Dict]:
    """`n_batches` batches of `batch` consecutive windows.  lonlat_to_pixel(lon, lat) -> (px, py)
    maps station coordinates to the INPUT grid (the reference indexes the 2x prediction with these
    input-grid indices, src/dataset.py:366-371 vs src/loss.py:130-140 - kept as is)."""
    days = n_batches * batch + window
    raw = make_days(days, channels, height, width, seed)
    rs = np.random.RandomState(seed)
    lon = rs.uniform(region[0], region[1], stations)
    lat = rs.uniform(region[2], region[3], stations)
    px, py = lonlat_to_pixel(lon, lat)
    run = np.zeros((days, stations), dtype=np.float64)
    for s in range(stations):
        y0, y1 = max(0, py[s] - 2), min(height, py[s] + 3)
        x0, x1 = max(0, px[s] - 2), min(width, px[s] + 3)
        run[:, s] = raw[:, 0, y0:y1, x0:x1].mean(axis=(1, 2)) + rs.randn(days) * 10.0
    run = np.maximum(run, 0)
    raw[:, 0][np.abs(raw[:, 0]) < 0.1] = 0.0
    mean = raw.mean(axis=(0, 2, 3), dtype=np.float64)
    std = raw.std(axis=(0, 2, 3), dtype=np.float64) + 1e-8
    norm = ((raw - mean[None, :, None, None]) / std[None, :, None, None]).astype(np.float32)
    run_n = ((run - run.mean()) / (run.std() + 1e-8)).astype(np.float32)
    pos = torch.from_numpy(np.stack([px, py], axis=1).astype(np.int64))
    out = []
    for b in range(n_batches):
        idx = [b * batch + i for i in range(batch)]
        images = np.stack([norm[i:i + window] for i in idx])
        target = np.stack([norm[i + window, 0:1] for i in idx])
        out.append({"images": torch.from_numpy(images), "output_image": torch.from_numpy(target),
                    "station_positions": [pos.clone() for _ in idx],
                    "stations": [torch.from_numpy(run_n[i + window].copy()) for i in idx]})
    return out


def make_dataset_inputs(days: int, channels: int = 10, height: int = 200, width: int = 200, stations: int = 64,
                        region=(100.0, 110.0, 25.0, 35.0), seed: int = 0):
    """Raw (unnormalised) inputs of a dataset directory in memory: frames [days,C,H,W] float32,
    date stems (YYYYMMDD from 2020-01-01, generate_dummy_data.py:205-206) and one DataFrame per
    station (timestamp, lon, lat, runoff = 5x5 mean of channel 0 + 10 N(0,1), clipped at 0)."""
    import datetime
    import pandas as pd
    raw = make_days(days, channels, height, width, seed)
    rs = np.random.RandomState(seed)
    lon = rs.uniform(region[0], region[1], stations)
    lat = rs.uniform(region[2], region[3], stations)
    px = np.clip(np.rint((lon - region[0]) / (region[1] - region[0]) * (width - 1)).astype(int), 0, width - 1)
    py = np.clip(np.rint((region[3] - lat) / (region[3] - region[2]) * (height - 1)).astype(int), 0, height - 1)
    start = datetime.date(2020, 1, 1)
    dates = [(start + datetime.timedelta(days=d)).strftime("%Y%m%d") for d in range(days)]
    frames = {}
    for s in range(stations):
        y0, y1 = max(0, py[s] - 2), min(height, py[s] + 3)
        x0, x1 = max(0, px[s] - 2), min(width, px[s] + 3)
        run = np.maximum(raw[:, 0, y0:y1, x0:x1].mean(axis=(1, 2)) + rs.randn(days) * 10.0, 0)
        frames[f"station_{s:03d}"] = pd.DataFrame({"timestamp": [int(d) for d in dates], "lon": lon[s], "lat": lat[s],
                                                   "runoff": run})
    return raw, dates, frames
