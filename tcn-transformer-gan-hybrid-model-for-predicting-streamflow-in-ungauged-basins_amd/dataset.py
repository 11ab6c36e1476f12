"""Device-resident counterpart of StreamflowDataset (src/dataset.py:13-511).

The reference loads T+1 .npy files and scans every station's DataFrame per sample on a loader
thread.  Here every date's frame is loaded once, kept in HBM as one float32 [D,C,H,W] tensor
(3650 x 10 x 200 x 200 is 5.8 GB), cleaned and normalised once by sf_dataset_transform, and a
sample is a view of T consecutive frames; station rows become dense [D,S] host tables built once.
A batch is one sf_gather_frames launch plus one small host-to-device copy of the packed station
rows.  Same constructor arguments, same sample dict, same statistics (float64) and the same
stats_cache.npz layout as the reference.

There is no CPU path: the frames live on the GPU and the arithmetic runs in the HIP library.
"""
from __future__ import annotations

from pathlib import Path
from typing import Dict, Iterator, List, Optional, Sequence

import numpy as np
import torch

from . import ops
from .functional import PackedStations


def _nearest_rows(n_in: int, n_out: int) -> np.ndarray:
    """Source index of F.interpolate(mode='nearest') (src/dataset.py:237-242): floor(dst * in/out),
    ratio in float32, clamped."""
    return np.minimum(np.floor(np.arange(n_out, dtype=np.float32) * np.float32(n_in / n_out)).astype(np.int64), n_in - 1)


class ResidentStreamflowDataset:
    """StreamflowDataset(image_dir, station_dir, config, normalize=True) with the frames resident
    on `device`.  `__getitem__` returns the reference's sample dict (images / output_image are
    device views, stations / station_positions host tensors); `batch` and `loader` serve
    collated batches in the layout of the reference's collate function (src/train.py:379-390)."""

    def __init__(self, image_dir: str, station_dir: str, config: Dict, normalize: bool = True, device: str = "cuda",
                 use_stats_cache: bool = True):
        import pandas as pd
        self.image_dir, self.station_dir = Path(image_dir), Path(station_dir)
        self._configure(config, normalize, device)
        # sorted by file name, mask files skipped (src/dataset.py:54-57)
        self.image_files = sorted(f for f in self.image_dir.iterdir()
                                  if f.suffix == ".npy" and not f.name.endswith("_mask.npy"))
        if not self.image_files:
            raise RuntimeError(f"No .npy files found in {image_dir}")
        dates = [f.stem for f in self.image_files]
        # directory order, unsorted, as the reference iterates it (src/dataset.py:88-91)
        stations = {p.stem: pd.read_csv(p) for p in self.station_dir.glob("*.csv")}

        def frames():
            for f in self.image_files:
                m = f.with_name(f.stem + "_mask.npy")
                yield np.load(f), (np.load(m) if m.exists() else None)

        self._build(frames(), len(dates), dates, stations, use_stats_cache)

    @classmethod
    def from_arrays(cls, frames: Sequence[np.ndarray], dates: Sequence[str], stations: Dict, config: Dict,
                    masks: Optional[Dict[int, np.ndarray]] = None, normalize: bool = True, device: str = "cuda"):
        """Same dataset from in-memory frames ([C,h,w] or [h,w] per date, in date order), optional
        {date index: [h,w] mask} and {station name: DataFrame with timestamp, lon, lat, runoff}."""
        self = cls.__new__(cls)
        self.image_dir = self.station_dir = None
        self._configure(config, normalize, device)
        self.image_files = [Path(str(d) + ".npy") for d in dates]
        masks = masks or {}
        self._build(((frames[d], masks.get(d)) for d in range(len(dates))), len(dates), [str(d) for d in dates],
                    stations, False)
        return self

    # ---------------------------------------------------------------- construction
    def _configure(self, config: Dict, normalize: bool, device: str):
        d = config.get("data", {})
        self.config, self.normalize = config, normalize
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("ResidentStreamflowDataset keeps its frames on the GPU: no CPU path exists")
        self.normalization_type = d.get("normalization_type", "z_score")
        self.window_size = d.get("window_size", 5)
        self.upscale_factor = d.get("upscale_factor", 2)
        self.region = d.get("region", [100, 110, 25, 35])
        self.resolution = d.get("resolution", 0.05)
        self.flow_threshold = d.get("flow_threshold", 0.1)
        size = d.get("image_size", [128, 128])
        self.input_height, self.input_width = (int(size), int(size)) if isinstance(size, int) else (int(size[0]), int(size[1]))
        self.image_mean = self.image_std = self.image_min = self.image_max = None
        self.runoff_mean = self.runoff_std = self.runoff_min = self.runoff_max = None

    def _resize(self, x: torch.Tensor) -> torch.Tensor:
        """[C,h,w] device f32 -> [C,H,W], bilinear align_corners=False (src/dataset.py:225-231)."""
        C, h, w = x.shape
        Cp = ops.roundup(C, 4)
        src = torch.zeros((1, h, w, Cp), device=self.device, dtype=torch.float32)
        src[0, :, :, :C] = x.permute(1, 2, 0)
        out = ops.bilinear_fwd(src, self.input_height, self.input_width, torch.float32)
        return out[0, :, :, :C].permute(2, 0, 1)

    def _build(self, frame_iter, D: int, dates: List[str], stations: Dict, use_cache: bool):
        H, W = self.input_height, self.input_width
        self.dates = dates
        self.stations = stations
        self.station_names = list(stations.keys())
        T = self.window_size
        # sliding windows (src/dataset.py:95-114)
        self.samples = [{"input_dates": dates[i:i + T], "output_date": dates[i + T],
                         "input_indices": list(range(i, i + T)), "output_index": i + T} for i in range(max(D - T, 0))]
        self.frames = None
        mask_rows, masks = [], []
        for d, (img, mask) in enumerate(frame_iter):
            x = torch.from_numpy(np.ascontiguousarray(img)).float()
            if x.dim() == 2:
                x = x.unsqueeze(0)
            if self.frames is None:
                self.num_channels = x.shape[0]
                self.frames = torch.empty((D, self.num_channels, H, W), device=self.device, dtype=torch.float32)
            x = x.to(self.device)
            self.frames[d] = self._resize(x) if (x.shape[1] != H or x.shape[2] != W) else x
            if mask is not None:
                m = np.asarray(mask).astype(np.float32)
                if m.shape != (H, W):
                    m = m[_nearest_rows(m.shape[0], H)][:, _nearest_rows(m.shape[1], W)]
                mask_rows.append(len(masks))
                masks.append(torch.from_numpy(np.ascontiguousarray(m)))
            else:
                mask_rows.append(-1)
        if masks:
            self._masks = torch.stack(masks).to(self.device)
            self._mask_row = torch.tensor(mask_rows, dtype=torch.int32, device=self.device)
        else:
            self._masks = self._mask_row = None

        part, count, bad0 = ops.dataset_plane_stats(self.frames, self._masks, self._mask_row)
        if self.normalize and not (use_cache and self._load_stats_from_cache()):
            self._compute_statistics(part.cpu().numpy(), count.cpu().numpy())
            if use_cache:
                self._save_stats_to_cache()
        a = b = None
        if self.normalize:
            if self.normalization_type == "minmax":
                a64, b64 = self.image_min, self.image_max - self.image_min
            else:
                a64, b64 = self.image_mean, self.image_std
            a = torch.from_numpy(np.ascontiguousarray(a64, dtype=np.float64)).to(self.device)
            b = torch.from_numpy(np.ascontiguousarray(b64, dtype=np.float64)).to(self.device)
        # a non-finite value in channel 0 stays in the reference's target but not in its inputs:
        # only then the targets need storage of their own
        self._own_targets = bool(int(bad0.sum().item()) > 0)
        self.targets = torch.empty((D, 1, H, W), device=self.device, dtype=torch.float32) if self._own_targets else None
        ops.dataset_transform(self.frames, self.targets, self._masks, self._mask_row, a, b, float(self.flow_threshold))
        self._build_station_tables()

    def _compute_statistics(self, part: np.ndarray, count: np.ndarray):
        """src/dataset.py:191-317 from the per-date partial sums: float64 accumulation in date order."""
        C = self.num_channels
        total = 0
        s, ss = np.zeros(C, np.float64), np.zeros(C, np.float64)
        mn, mx = np.full(C, np.inf), np.full(C, -np.inf)
        for d in range(part.shape[0]):
            if count[d] == 0:
                continue
            total += int(count[d])
            s += part[d, :, 0]
            ss += part[d, :, 1]
            mn = np.minimum(mn, part[d, :, 2])
            mx = np.maximum(mx, part[d, :, 3])
        if total == 0:
            raise RuntimeError("No valid pixels found to compute statistics.")
        self.image_mean = s / total
        self.image_std = np.sqrt(np.maximum(ss / total - self.image_mean ** 2, 0.0))
        self.image_std[self.image_std == 0] = 1.0
        self.image_min, self.image_max = mn, mx
        self.image_max[self.image_max == self.image_min] += 1.0
        n, rs, rss, rmin, rmax = 0, 0.0, 0.0, float("inf"), float("-inf")
        for name, df in self.stations.items():
            if "runoff" not in df.columns:
                continue
            arr = df["runoff"].values.astype(np.float64)
            if not np.all(np.isfinite(arr)):
                arr = np.nan_to_num(arr)
            n += arr.size
            rs += float(arr.sum())
            rss += float((arr ** 2).sum())
            if arr.size:
                rmin, rmax = min(rmin, float(arr.min())), max(rmax, float(arr.max()))
        if n:
            self.runoff_mean = rs / n
            self.runoff_std = float(np.sqrt(max(rss / n - self.runoff_mean ** 2, 0.0))) or 1.0
            self.runoff_min, self.runoff_max = rmin, (rmin + 1.0 if rmax == rmin else rmax)
        else:
            self.runoff_mean, self.runoff_std, self.runoff_min, self.runoff_max = 0.0, 1.0, 0.0, 1.0

    # stats cache: the reference's file (src/dataset.py:122-189), read without unpickling
    def _stats_cache_path(self) -> Path:
        out = Path(self.config.get("data", {}).get("output_dir", "./output"))
        out.mkdir(parents=True, exist_ok=True)
        return out / "stats_cache.npz"

    def _load_stats_from_cache(self) -> bool:
        path = self._stats_cache_path()
        if not path.exists():
            return False
        try:
            data = np.load(path, allow_pickle=False)
            if [str(x) for x in data["image_files"]] != [str(p) for p in self.image_files]:
                return False
            if "image_size" in data.files:
                h, w = data["image_size"]
                if int(h) != self.input_height or int(w) != self.input_width:
                    return False
            if int(data["num_channels"]) != self.num_channels:
                return False
            self.image_mean, self.image_std = data["image_mean"], data["image_std"]
            self.image_min, self.image_max = data["image_min"], data["image_max"]
            self.runoff_mean, self.runoff_std = float(data["runoff_mean"]), float(data["runoff_std"])
            self.runoff_min, self.runoff_max = float(data["runoff_min"]), float(data["runoff_max"])
            return True
        except Exception as e:  # an unreadable cache is recomputed, as in the reference
            print(f"Warning: failed to load stats cache from {path}: {e}")
            return False

    def _save_stats_to_cache(self):
        np.savez(self._stats_cache_path(), image_files=[str(p) for p in self.image_files],
                 image_size=[self.input_height, self.input_width], num_channels=self.num_channels,
                 image_mean=self.image_mean, image_std=self.image_std, image_min=self.image_min,
                 image_max=self.image_max, runoff_mean=self.runoff_mean, runoff_std=self.runoff_std,
                 runoff_min=self.runoff_min, runoff_max=self.runoff_max)

    def _build_station_tables(self):
        """_get_station_target (src/dataset.py:319-350) for every date at once: per station the
        first row whose timestamp string equals the date; pixel position from that row's lon/lat
        (sf_lonlat_to_pixel, float64); runoff normalised in float64, stored as float32."""
        import pandas as pd
        D, S = len(self.dates), len(self.station_names)
        present = np.zeros((D, S), bool)
        lon, lat, run = np.zeros((D, S)), np.zeros((D, S)), np.zeros((D, S))
        index = pd.Index(self.dates)
        for s, name in enumerate(self.station_names):
            df = self.stations[name]
            ts = df["timestamp"].astype(str)
            first = ~ts.duplicated(keep="first")
            rows = pd.Index(ts[first]).get_indexer(index)          # row (among first occurrences) per date, -1 if none
            hit = rows >= 0
            sub = df[first.values]
            present[hit, s] = True
            lon[hit, s] = sub["lon"].values.astype(np.float64)[rows[hit]]
            lat[hit, s] = sub["lat"].values.astype(np.float64)[rows[hit]]
            run[hit, s] = sub["runoff"].values.astype(np.float64)[rows[hit]]
        self.station_present = present
        if S and present.any():
            px, py = ops.lonlat_to_pixel(torch.from_numpy(lon.reshape(-1)).to(self.device),
                                         torch.from_numpy(lat.reshape(-1)).to(self.device), self.region,
                                         self.input_width, self.input_height)
            pos = torch.stack([px, py], dim=1).cpu().numpy().reshape(D, S, 2)
        else:
            pos = np.zeros((D, S, 2), np.int32)
        if self.normalize:
            if self.normalization_type == "minmax":
                run = (run - self.runoff_min) / (self.runoff_max - self.runoff_min)
            else:
                run = (run - self.runoff_mean) / self.runoff_std
        self.station_pos = pos.astype(np.int32)
        self.station_runoff = run.astype(np.float32)
        self.station_count = present.sum(axis=1).astype(np.int64)

    # ---------------------------------------------------------------- samples
    def __len__(self) -> int:
        return len(self.samples)

    def _target_view(self, o: int) -> torch.Tensor:
        return self.targets[o] if self._own_targets else self.frames[o, 0:1]

    def _stations_of(self, o: int):
        sel = self.station_present[o]
        return self.station_pos[o][sel], self.station_runoff[o][sel]

    def __getitem__(self, idx: int) -> Dict:
        if idx < 0:
            idx += len(self)
        if not 0 <= idx < len(self):
            raise IndexError(idx)
        T = self.window_size
        o = idx + T
        pos, run = self._stations_of(o)
        return {"images": self.frames[idx:idx + T], "output_image": self._target_view(o),
                "stations": torch.from_numpy(run.copy()),
                "station_positions": torch.from_numpy(pos.astype(np.int64)).reshape(-1, 2) if pos.size
                else torch.zeros(0, dtype=torch.int64),
                "date": self.dates[o]}

    def sample_station_names(self, idx: int) -> List[str]:
        o = idx + self.window_size
        return [n for n, p in zip(self.station_names, self.station_present[o]) if p]

    def batch(self, indices: Sequence[int]) -> Dict:
        """Collated batch of the given window indices: images [B,T,C,H,W] and output_image
        [B,1,H,W] on the device (one gather launch each), the station rows packed for the loss
        kernel (both `stations` and `station_positions` hold the same PackedStations), dates."""
        idx = np.asarray(indices, dtype=np.int64).reshape(-1)
        B, T = idx.size, self.window_size
        if B == 0 or idx.min() < 0 or idx.max() >= len(self):
            raise IndexError("window index out of range")
        D, C, H, W = self.frames.shape
        rows = np.concatenate([(idx[:, None] + np.arange(T)[None]).reshape(-1), idx + T])
        rows_d = ops.upload(torch.from_numpy(rows), self.device)
        images = torch.empty((B, T, C, H, W), device=self.device, dtype=torch.float32)
        target = torch.empty((B, 1, H, W), device=self.device, dtype=torch.float32)
        ops.gather_frames(self.frames, C * H * W, D, rows_d[:B * T], C * H * W, images)
        if self._own_targets:
            ops.gather_frames(self.targets, H * W, D, rows_d[B * T:], H * W, target)
        else:
            ops.gather_frames(self.frames, C * H * W, D, rows_d[B * T:], H * W, target)
        o = idx + T
        sel = self.station_present[o]                      # [B,S]
        counts = sel.sum(axis=1)
        packed = PackedStations.from_host(self.station_pos[o][sel], self.station_runoff[o][sel], counts, self.device)
        return {"images": images, "output_image": target, "stations": packed, "station_positions": packed,
                "dates": [self.dates[k] for k in o]}

    def loader(self, batch_size: int, shuffle: bool = True, seed: int = 0, epoch: int = 0, rank: int = 0,
               world: int = 1, drop_last: bool = False) -> Iterator[Dict]:
        """Batches of one epoch.  With world > 1 every rank walks a disjoint, equally long slice
        of the same permutation (the tail that does not divide is dropped so that all ranks take
        the same number of steps)."""
        for ind in window_shards(len(self), batch_size, shuffle, seed, epoch, rank, world, drop_last):
            yield self.batch(ind)


def window_shards(n: int, batch_size: int, shuffle: bool, seed: int, epoch: int, rank: int, world: int,
                  drop_last: bool) -> List[np.ndarray]:
    """Index batches of one rank for one epoch: a seeded permutation (the same on every rank) cut
    to a multiple of `world`, rank r taking elements r, r+world, ..."""
    if shuffle:
        g = torch.Generator()
        g.manual_seed(seed + epoch)
        order = torch.randperm(n, generator=g).numpy()
    else:
        order = np.arange(n)
    per = n // world
    mine = order[: per * world][rank::world]
    out = [mine[i:i + batch_size] for i in range(0, per, batch_size)]
    if drop_last and out and len(out[-1]) < batch_size:
        out.pop()
    return out
