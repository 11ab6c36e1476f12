"""Deterministic tensors shared by the golden-vector generator and the tests.

Weights and inputs of every golden case are produced by formula from a name, so the
fixtures only have to store the reference's OUTPUTS.  Nothing here imports the reference.
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Tuple

import numpy as np
import torch


def gold_array(name: str, shape, scale: float = 1.0, dtype=np.float32) -> np.ndarray:
    rs = np.random.RandomState(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    return (rs.standard_normal(tuple(shape)) * scale).astype(dtype)


def gold_tensor(name: str, shape, scale: float = 1.0) -> torch.Tensor:
    return torch.from_numpy(gold_array(name, shape, scale))


def gold_state(shapes: Dict[str, Tuple[int, ...]], tag: str, keep: Dict[str, torch.Tensor] = None
               ) -> Dict[str, torch.Tensor]:
    """Deterministic state_dict for the given key -> shape map.

    Conv / linear / LSTM weights ~ N(0, 1/fan_in); biases ~ N(0, 0.1^2); norm weights
    1 + N(0, 0.1^2); running_var 1 + |N(0, 0.2^2)|; counters 0; positional-encoding buffers are
    taken from `keep` (they are a closed formula of the module)."""
    out = {}
    for k, shp in shapes.items():
        shp = tuple(shp)
        nm = f"{tag}/{k}"
        if k.endswith("num_batches_tracked"):
            out[k] = torch.zeros((), dtype=torch.int64)
        elif k.endswith("pos_encoder.pe"):
            out[k] = keep[k].clone()
        elif k.endswith("running_mean"):
            out[k] = gold_tensor(nm, shp, 0.1)
        elif k.endswith("running_var"):
            out[k] = 1.0 + gold_tensor(nm, shp, 0.2).abs()
        elif len(shp) == 1 and (".norm" in k or _is_bn_weight(k, shapes)) and k.endswith("weight"):
            out[k] = 1.0 + gold_tensor(nm, shp, 0.1)
        elif len(shp) == 1:
            out[k] = gold_tensor(nm, shp, 0.1)
        else:
            fan_in = int(np.prod(shp[1:]))
            out[k] = gold_tensor(nm, shp, 1.0 / np.sqrt(fan_in))
    return out


def _is_bn_weight(k: str, shapes) -> bool:
    return (k[: -len("weight")] + "running_mean") in shapes


def shapes_of(sd: Dict[str, torch.Tensor]) -> Dict[str, Tuple[int, ...]]:
    return {k: tuple(v.shape) for k, v in sd.items()}


def checksum(sd: Dict[str, torch.Tensor], keys: Iterable[str] = None) -> Dict[str, np.ndarray]:
    """Per-key [sum, abs-sum] in float64."""
    out = {}
    for k in (keys if keys is not None else sd.keys()):
        v = sd[k].detach().to(torch.float64)
        out[k] = np.array([float(v.sum()), float(v.abs().sum())])
    return out


def update_projection(after: Dict[str, torch.Tensor], before: Dict[str, torch.Tensor], keys: Iterable[str]
                      ) -> Dict[str, np.ndarray]:
    """Per key [sum_k r_k * (after - before)_k, ||after - before||_2] in float64, r a fixed
    pseudo-random +-1 / normal direction made from the key's name.  Unlike an abs-sum checksum of
    the parameter itself, a signed projection of the UPDATE moves by the full size of any error in
    it: a missing, doubled or mis-signed optimiser step on a tensor changes it by O(1) of itself."""
    out = {}
    for k in keys:
        d = (after[k].detach().to(torch.float64) - before[k].detach().to(torch.float64)).cpu().numpy()
        r = gold_array("proj/" + k, d.shape, 1.0, np.float64)
        out[k] = np.array([float((r * d).sum()), float(np.sqrt((d * d).sum()))])
    return out


def update_signs(after: Dict[str, torch.Tensor], before: Dict[str, torch.Tensor], keys: Iterable[str]) -> np.ndarray:
    """[after > before] of every entry of the given tensors, concatenated in key order and packed
    eight to a byte.  Adam's FIRST step moves every entry by exactly lr against the sign of its
    gradient, so for a network that has taken one step these bits are the whole update."""
    bits = [(after[k].detach().to(torch.float64) > before[k].detach().to(torch.float64)).cpu().numpy().reshape(-1)
            for k in keys]
    return np.packbits(np.concatenate(bits))


# ---- pixel-map sweep: random points, exact .5 ties, edges, outside-region points
PIXEL_CASES = [
    ((100.0, 110.0, 25.0, 35.0), 128, 128),
    ((100.0, 110.0, 25.0, 35.0), 200, 200),
    ((100.0, 110.0, 25.0, 35.0), 600, 400),
    ((90.0, 150.0, -15.0, 25.0), 128, 128),
    ((90.0, 150.0, -15.0, 25.0), 200, 200),
    ((90.0, 150.0, -15.0, 25.0), 600, 400),
]


def pixel_sweep(region, width: int, height: int, n_random: int = 3000):
    lon_min, lon_max, lat_min, lat_max = region
    rs = np.random.RandomState(width * 1000 + height + int(lon_min))
    lon = rs.uniform(lon_min - 1.0, lon_max + 1.0, n_random)
    lat = rs.uniform(lat_min - 1.0, lat_max + 1.0, n_random)
    # exact half-pixel ties: x_ratio*(W-1) = k + 0.5
    k = np.arange(0, width - 1, dtype=np.float64)
    tie_lon = lon_min + (k + 0.5) / (width - 1) * (lon_max - lon_min)
    j = np.arange(0, height - 1, dtype=np.float64)
    tie_lat = lat_max - (j + 0.5) / (height - 1) * (lat_max - lat_min)
    n = min(len(tie_lon), len(tie_lat))
    edge_lon = np.array([lon_min, lon_max, lon_min, lon_max, (lon_min + lon_max) / 2, lon_min - 50, lon_max + 50])
    edge_lat = np.array([lat_min, lat_max, lat_max, lat_min, (lat_min + lat_max) / 2, lat_min - 50, lat_max + 50])
    lon = np.concatenate([lon, tie_lon[:n], edge_lon, tie_lon[:n][::-1]])
    lat = np.concatenate([lat, tie_lat[:n], edge_lat, tie_lat[:n]])
    return lon.astype(np.float64), lat.astype(np.float64)


DATASET_CASES = {
    "A": {"window_size": 5, "normalization_type": "z_score"},
    "B": {"window_size": 3, "normalization_type": "minmax"},
}
DATASET_STATIONS = ["st_b", "st_a", "gauge10", "x9"]


def dataset_case(g, tag):
    """Raw inputs of dataset case `tag` from the dataset.npz fixture: frames [D,C,h,w], masks
    {date index: [h,w] uint8}, dates (file stems), stations {name: (timestamps as str, lon, lat,
    runoff)} and the config the reference ran with."""
    frames = g[f"{tag}_frames"]
    masks = {int(d): g[f"{tag}_masks"][i] for i, d in enumerate(g[f"{tag}_mask_idx"])}
    dates = [f"202001{d + 1:02d}" for d in range(frames.shape[0])]
    stations = {}
    for n in DATASET_STATIONS:
        a = g[f"{tag}_st_{n}"]
        stations[n] = ([str(int(t)) for t in a[0]], a[1], a[2], a[3])
    cfg = {"data": {"upscale_factor": 2, "region": [100, 110, 25, 35], "flow_threshold": 0.1, "image_size": [20, 24],
                    **DATASET_CASES[tag]}}
    return frames, masks, dates, stations, cfg


def write_dataset_case(root, frames, masks, dates, stations):
    """The directory layout StreamflowDataset reads (src/dataset.py:54-57,88-91)."""
    import os
    import pandas as pd
    os.makedirs(os.path.join(root, "images"))
    os.makedirs(os.path.join(root, "stations"))
    for d, name in enumerate(dates):
        np.save(os.path.join(root, "images", name + ".npy"), frames[d])
        if d in masks:
            np.save(os.path.join(root, "images", name + "_mask.npy"), masks[d])
    for n, (ts, lon, lat, run) in stations.items():
        pd.DataFrame({"timestamp": [int(t) for t in ts], "lon": lon, "lat": lat, "runoff": run}).to_csv(
            os.path.join(root, "stations", n + ".csv"), index=False)
