"""Inference / evaluation driver: the counterpart of Evaluator (src/evaluate.py:89-335) and of
calculate_metrics (src/loss.py:239-344) for the HIP model.

Same outputs as the reference (image-level RMSE / MAE over all pixels, per-date SSIM / PSNR /
correlation averaged over dates, k x k station means and their metrics, the optional pred_*.npy /
target_*.npy / stations_eval.csv / evaluation_metrics.csv files).  What differs is the shape of
the work: dates are independent, so they are forwarded in batches instead of one by one, and a
list of dates can be sharded over ranks (one process per GPU); every shard returns order-free
partial results that `merge_partials` combines into exactly the unsharded answer.  The metric
arithmetic stays on the host in NumPy / SciPy, as in the reference, for ordinary datasets; for a
ResidentStreamflowDataset (frames already in HBM) the per-date image metrics come from
sf_date_image_metrics on the device, consecutive dates share their encoded frames
(StreamflowPredictionModel.forward_windows) and ranks take contiguous blocks of dates.
"""
from __future__ import annotations

import csv
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

from . import ops
from .modules import StreamflowPredictionModel


# ------------------------------------------------------------------------------ metrics (host)
def calculate_ssim(pred: np.ndarray, target: np.ndarray, data_range: Optional[float] = None) -> float:
    """Mean SSIM with an 11 x 11 uniform window, reflect borders, K1 = 0.01, K2 = 0.03
    (src/evaluate.py:17-60)."""
    from scipy.ndimage import uniform_filter
    if data_range is None:
        data_range = max(target.max() - target.min(), pred.max() - pred.min())
    if data_range == 0:
        return 1.0 if np.allclose(pred, target) else 0.0
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    mu_p, mu_t = uniform_filter(pred, size=11), uniform_filter(target, size=11)
    var_p = uniform_filter(pred ** 2, size=11) - mu_p ** 2
    var_t = uniform_filter(target ** 2, size=11) - mu_t ** 2
    cov = uniform_filter(pred * target, size=11) - mu_p * mu_t
    ssim = ((2 * mu_p * mu_t + c1) * (2 * cov + c2)) / ((mu_p ** 2 + mu_t ** 2 + c1) * (var_p + var_t + c2))
    return float(np.mean(ssim))


def calculate_psnr(pred: np.ndarray, target: np.ndarray, data_range: Optional[float] = None) -> float:
    """20 log10(range / rmse) (src/evaluate.py:63-86)."""
    if data_range is None:
        data_range = max(target.max() - target.min(), pred.max() - pred.min())
    if data_range == 0:
        return float("inf") if np.allclose(pred, target) else 0.0
    mse = np.mean((pred - target) ** 2)
    if mse == 0:
        return float("inf")
    return float(20 * np.log10(data_range / np.sqrt(mse)))


def calculate_metrics(predictions, targets) -> Dict[str, float]:
    """RMSE, MAE, R2, NSE, MAPE (%), KGE of predictions against targets (src/loss.py:239-344;
    non-finite values count as zero).  A 4-D [B,C,h,w] or 3-D [C,h,w] target whose shape differs
    from the prediction's is first resized bilinearly (align_corners=False) to the prediction's
    last two dimensions, as src/loss.py:254-271 does with F.interpolate; here that resize is the
    HIP kernel, so this one case needs the GPU.  Any other shape mismatch raises ValueError (the
    reference fails in NumPy's broadcast at the same point)."""
    def flat(a):
        if isinstance(a, torch.Tensor):
            a = a.detach().cpu().numpy()
        return np.nan_to_num(np.asarray(a).reshape(-1), nan=0.0, posinf=0.0, neginf=0.0)

    ps, ts = tuple(np.shape(predictions)), tuple(np.shape(targets))
    if ps != ts and len(ts) in (3, 4) and len(ps) >= 2:
        t4 = torch.as_tensor(np.asarray(targets) if not isinstance(targets, torch.Tensor) else targets).detach().float()
        t4 = t4.reshape((-1, 1) + ts[-2:])                 # every (batch, channel) plane is resized on its own
        targets = resize_bilinear(t4, int(ps[-2]), int(ps[-1]), "cuda").reshape(ts[:-2] + (int(ps[-2]), int(ps[-1])))
    p, t = flat(predictions), flat(targets)
    zero = {"RMSE": 0.0, "MAE": 0.0, "R2": 0.0, "NSE": 0.0, "MAPE": 0.0, "KGE": 0.0}
    if p.size and t.size and p.shape != t.shape:
        raise ValueError(f"calculate_metrics: {ps} against {ts} cannot be compared (only 3-D / 4-D targets are resized)")
    if p.size == 0 or t.size == 0:
        return zero
    err = p - t
    ss_res, ss_tot = float((err ** 2).sum()), float(((t - t.mean()) ** 2).sum())
    nse = 1.0 - ss_res / ss_tot if ss_tot > 0 else 0.0
    big = np.abs(t) > 1e-6
    mape = float(np.abs(err[big] / t[big]).mean() * 100.0) if big.any() else 0.0
    kge = 0.0
    sp, st, mt = float(p.std()), float(t.std()), float(t.mean())
    if sp > 0 and st > 0 and mt != 0.0:
        r = float(np.corrcoef(p, t)[0, 1])
        kge = 1.0 - float(np.sqrt((r - 1.0) ** 2 + (sp / st - 1.0) ** 2 + (float(p.mean()) / mt - 1.0) ** 2))
    return {"RMSE": float((err ** 2).mean() ** 0.5), "MAE": float(np.abs(err).mean()), "R2": float(nse),
            "NSE": float(nse), "MAPE": mape, "KGE": float(kge)}


# ------------------------------------------------------------------------------ partial results
def _new_partial() -> Dict:
    return {"sse": 0.0, "sae": 0.0, "npx": 0, "per_date": [], "stations": []}


def merge_partials(partials: Sequence[Dict]) -> Dict:
    """Partial results of disjoint date shards -> the metrics of the whole list (the per-date
    entries carry their dataset index, so the order of the shards does not matter)."""
    sse = sum(float(p["sse"]) for p in partials)
    sae = sum(float(p["sae"]) for p in partials)
    npx = sum(int(p["npx"]) for p in partials)
    per_date = sorted((e for p in partials for e in p["per_date"]), key=lambda e: e[0])
    stations = sorted((e for p in partials for e in p["stations"]), key=lambda e: e[0])
    ssims = [e[1] for e in per_date]
    psnrs = [e[2] for e in per_date if e[2] is not None]
    corrs = [e[3] for e in per_date if e[3] is not None]
    out = {"RMSE": float(np.sqrt(sse / npx)) if npx else 0.0, "MAE": float(sae / npx) if npx else 0.0,
           "SSIM": float(np.mean(ssims)) if ssims else 0.0, "PSNR": float(np.mean(psnrs)) if psnrs else 0.0,
           "Correlation": float(np.mean(corrs)) if corrs else 0.0}
    if stations:
        arr = np.array([(e[4], e[5]) for e in stations], dtype=np.float64)
        out["station_metrics"] = calculate_metrics(arr[:, 1], arr[:, 0])
        out["station_rows"] = [(e[1], e[2], e[3], e[4], e[5]) for e in stations]
    return out


def resize_bilinear_device(target: torch.Tensor, OH: int, OW: int) -> torch.Tensor:
    """resize_bilinear for a device tensor [B,1,h,w]; the result stays on the device."""
    B, _, h, w = target.shape
    Bp = ops.roundup(B, 4)
    src = torch.zeros((1, h, w, Bp), device=target.device, dtype=torch.float32)
    src[0, :, :, :B] = target.float()[:, 0].permute(1, 2, 0)
    out = ops.bilinear_fwd(src, OH, OW, torch.float32)
    return out[0, :, :, :B].permute(2, 0, 1).unsqueeze(1).contiguous()


def per_date_from_moments(i: int, m: Sequence[float], npx: int):
    """(index, SSIM, PSNR or None, correlation or None) of one date from a row of
    sf_date_image_metrics, with the reference's special cases (src/evaluate.py:30-35,76-86,
    218-230): zero data range, zero error, constant maps."""
    sse, sae, sp, st, spp, stt, spt, pmin, pmax, tmin, tmax, ssim_sum, rng = [float(v) for v in m]
    same = abs(pmin - tmin) <= 1e-8 + 1e-5 * abs(tmin)      # np.allclose of two constant maps
    if rng == 0:
        ssim, psnr = (1.0 if same else 0.0), (float("inf") if same else 0.0)
    else:
        ssim = ssim_sum / npx
        mse = sse / npx
        psnr = float("inf") if mse == 0 else float(20 * np.log10(rng / np.sqrt(mse)))
    mp, mt = sp / npx, st / npx
    vp, vt = max(spp / npx - mp * mp, 0.0), max(stt / npx - mt * mt, 0.0)
    corr = None
    if np.sqrt(vp) > 1e-10 and np.sqrt(vt) > 1e-10:
        c = (spt / npx - mp * mt) / np.sqrt(vp * vt)
        corr = None if np.isnan(c) else float(c)
    return (i, float(ssim), None if np.isinf(psnr) else psnr, corr)


def resize_bilinear(target: torch.Tensor, OH: int, OW: int, device) -> torch.Tensor:
    """Bilinear (align_corners=False) resize of [B,1,h,w] on the GPU (the kernel of
    F.interpolate's counterpart, sf_bilinear_fwd): the batch rides in the channel dimension of
    one NHWC image.  Returns a host tensor [B,1,OH,OW]."""
    B, _, h, w = target.shape
    Bp = ops.roundup(B, 4)
    src = torch.zeros((1, h, w, Bp), device=device, dtype=torch.float32)
    src[0, :, :, :B] = target.to(device).float()[:, 0].permute(1, 2, 0)
    out = ops.bilinear_fwd(src, OH, OW, torch.float32)
    return out[0, :, :, :B].permute(2, 0, 1).unsqueeze(1).contiguous().cpu()


class Evaluator:
    """Evaluator(config, model_path) as in the reference; `model` may be passed instead of a
    checkpoint path.  The model runs in eval mode on the GPU (f32 unless the caller wraps the
    calls in torch.autocast)."""

    def __init__(self, config: Dict, model_path: Optional[str] = None, model: Optional[torch.nn.Module] = None,
                 device: str = "cuda"):
        self.config = config
        self.device = torch.device(device)
        if model is None:
            model = StreamflowPredictionModel(config)
            if model_path is not None:
                model.load_state_dict(torch.load(model_path, map_location="cpu", weights_only=True))
        self.model = model.to(self.device)
        self.model.eval()

    # -- forward ---------------------------------------------------------------------------
    def predict_batch(self, images: torch.Tensor) -> torch.Tensor:
        """[B,T,C,H,W] -> [B,1,H_out,W_out] on the host (src/evaluate.py:129-147)."""
        with torch.no_grad():
            return self.model(images.to(self.device, non_blocking=True)).float().cpu()

    def _resize_targets(self, target: torch.Tensor, OH: int, OW: int) -> torch.Tensor:
        return resize_bilinear(target, OH, OW, self.device)

    # -- evaluation ------------------------------------------------------------------------
    def _date_metrics(self, i: int, s: Dict, p2: np.ndarray, t2: np.ndarray, half: int, pred_dir: Optional[str]):
        """Host work of one date (src/evaluate.py:205-262): error sums, SSIM / PSNR / correlation,
        the k x k station means, the optional .npy files."""
        p, t = p2.reshape(-1), t2.reshape(-1)
        sse, sae = float(((p - t) ** 2).sum()), float(np.abs(p - t).sum())
        ps = calculate_psnr(p2, t2)
        corr = None
        if p.std() > 1e-10 and t.std() > 1e-10:
            c = np.corrcoef(p, t)[0, 1]
            corr = None if np.isnan(c) else float(c)
        per_date = (i, calculate_ssim(p2, t2), None if np.isinf(ps) else ps, corr)
        date = s.get("date", f"{i:06d}")
        if pred_dir is not None:
            np.save(os.path.join(pred_dir, f"pred_{date}.npy"), p2)
            np.save(os.path.join(pred_dir, f"target_{date}.npy"), t2)
        stations = []
        for j, (pos, r) in enumerate(zip(s["station_positions"], s["stations"])):
            px, py = int(pos[0]), int(pos[1])
            y0, y1 = max(0, py - half), min(p2.shape[0], py + half + 1)
            x0, x1 = max(0, px - half), min(p2.shape[1], px + half + 1)
            stations.append(((i, j), date, px, py, float(r), float(p2[y0:y1, x0:x1].mean())))
        return sse, sae, p.size, per_date, stations

    def _station_rows(self, i: int, date: str, pos, runs, p2: np.ndarray, half: int):
        rows = []
        for j, (q, r) in enumerate(zip(pos, runs)):
            px, py = int(q[0]), int(q[1])
            y0, y1 = max(0, py - half), min(p2.shape[0], py + half + 1)
            x0, x1 = max(0, px - half), min(p2.shape[1], px + half + 1)
            rows.append(((i, j), date, px, py, float(r), float(p2[y0:y1, x0:x1].mean())))
        return rows

    def evaluate_partial_resident(self, dataset, indices: Sequence[int], batch_size: int = 16,
                                  pred_dir: Optional[str] = None) -> Dict:
        """evaluate_partial for a ResidentStreamflowDataset: windows are views of the resident
        frames, a run of consecutive dates encodes each frame once, targets are resized and the
        image metrics reduced on the device (sf_date_image_metrics); the host keeps the k x k
        station means and the optional .npy files."""
        part = _new_partial()
        half = self.config.get("stations", {}).get("kernel_size", 3) // 2
        idx = list(indices)
        T = dataset.window_size
        pending = None

        def finish(job):
            chunk, preds_h, tg_h, mom_h = job
            npx = preds_h.shape[-1] * preds_h.shape[-2]
            for k, i in enumerate(chunk):
                m = mom_h[k]
                part["sse"] += float(m[0])
                part["sae"] += float(m[1])
                part["npx"] += npx
                part["per_date"].append(per_date_from_moments(i, m, npx))
                o = i + T
                date = dataset.dates[o]
                p2 = preds_h[k, 0]
                if pred_dir is not None:
                    np.save(os.path.join(pred_dir, f"pred_{date}.npy"), p2)
                    np.save(os.path.join(pred_dir, f"target_{date}.npy"), tg_h[k, 0])
                sel = dataset.station_present[o]
                part["stations"] += self._station_rows(i, date, dataset.station_pos[o][sel],
                                                       dataset.station_runoff[o][sel], p2, half)

        for b0 in range(0, len(idx), batch_size):
            chunk = idx[b0:b0 + batch_size]
            with torch.no_grad():
                if all(b - a == 1 for a, b in zip(chunk, chunk[1:])):
                    starts = torch.arange(len(chunk), device=self.device)
                    preds = self.model.forward_windows(dataset.frames[chunk[0]:chunk[-1] + T], starts, T).float()
                    tg = torch.stack([dataset._target_view(i + T) for i in chunk]) if dataset._own_targets \
                        else dataset.frames[chunk[0] + T:chunk[-1] + T + 1, 0:1].contiguous()
                else:
                    b = dataset.batch(chunk)
                    preds, tg = self.model(b["images"]).float(), b["output_image"]
                if tg.shape != preds.shape:
                    tg = resize_bilinear_device(tg, preds.shape[2], preds.shape[3])
                mom = ops.date_image_metrics(preds.contiguous(), tg.contiguous())
            # the previous chunk's host work runs while this chunk's kernels do
            if pending is not None:
                finish(pending)
            pending = (chunk, preds.cpu().numpy(), tg.cpu().numpy() if pred_dir is not None else None, mom.cpu().numpy())
        if pending is not None:
            finish(pending)
        return part

    def evaluate_partial(self, dataset: Sequence[Dict], indices: Sequence[int], batch_size: int = 16,
                         pred_dir: Optional[str] = None, workers: Optional[int] = None) -> Dict:
        """Metrics contributions of dataset[i] for i in indices (src/evaluate.py:186-268).  The
        per-date host metrics (SciPy filters dominate) run on a thread pool while the GPU forwards
        the next batch; results are collected in index order, so the sums do not depend on timing."""
        from concurrent.futures import ThreadPoolExecutor
        part = _new_partial()
        half = self.config.get("stations", {}).get("kernel_size", 3) // 2
        idx = list(indices)
        if workers is None:
            workers = max(1, min(16, os.cpu_count() or 1))
        futures = []
        with ThreadPoolExecutor(max_workers=workers) as pool:
            for b0 in range(0, len(idx), batch_size):
                chunk = idx[b0:b0 + batch_size]
                samples = [dataset[i] for i in chunk]
                preds = self.predict_batch(torch.stack([s["images"] for s in samples]))
                tgts = torch.stack([s["output_image"] for s in samples]).float()
                if tgts.shape != preds.shape:
                    tgts = self._resize_targets(tgts, preds.shape[2], preds.shape[3])
                for i, s, pr, tg in zip(chunk, samples, preds.numpy(), tgts.numpy()):
                    futures.append(pool.submit(self._date_metrics, i, s, pr[0], tg[0], half, pred_dir))
            for f in futures:
                sse, sae, npx, per_date, stations = f.result()
                part["sse"] += sse
                part["sae"] += sae
                part["npx"] += npx
                part["per_date"].append(per_date)
                part["stations"] += stations
        return part

    def evaluate_dataset(self, dataset: Sequence[Dict], output_dir: Optional[str] = None, batch_size: int = 16,
                         shard: Optional[Tuple[int, int]] = None) -> Dict:
        """Evaluator.evaluate_dataset (src/evaluate.py:149-335).  `shard=(rank, world)` evaluates
        dates rank, rank+world, ...; with an initialised process group the shards' partial results
        are gathered and every rank returns the metrics of the whole dataset (rank 0 writes the
        CSV files)."""
        rank, world = shard if shard is not None else (0, 1)
        pred_dir = None
        if output_dir is not None:
            pred_dir = os.path.join(output_dir, "predictions")
            os.makedirs(pred_dir, exist_ok=True)
        from .dataset import ResidentStreamflowDataset
        if isinstance(dataset, ResidentStreamflowDataset):
            # contiguous blocks of dates per rank: neighbours share window - 1 encoded frames
            n = len(dataset)
            lo, hi = rank * n // world, (rank + 1) * n // world
            part = self.evaluate_partial_resident(dataset, range(lo, hi), batch_size, pred_dir)
        else:
            part = self.evaluate_partial(dataset, range(rank, len(dataset), world), batch_size, pred_dir)
        parts: List[Dict] = [part]
        if world > 1 and dist.is_available() and dist.is_initialized():
            parts = [None] * dist.get_world_size()
            dist.all_gather_object(parts, part)
        elif world > 1:
            return {"partial": part}
        metrics = merge_partials(parts)
        if output_dir is not None and rank == 0:
            self._write_csv(output_dir, metrics)
        return metrics

    @staticmethod
    def _write_csv(output_dir: str, metrics: Dict):
        rows = metrics.get("station_rows", [])
        if not rows:
            return
        with open(os.path.join(output_dir, "stations_eval.csv"), "w", newline="") as f:
            wr = csv.writer(f)
            wr.writerow(["date", "position_x", "position_y", "actual_runoff", "predicted_runoff"])
            wr.writerows(rows)
        flat = {k: v for k, v in metrics.items() if k not in ("station_metrics", "station_rows")}
        flat.update({f"station_{k}": v for k, v in metrics["station_metrics"].items()})
        with open(os.path.join(output_dir, "evaluation_metrics.csv"), "w", newline="") as f:
            wr = csv.writer(f)
            wr.writerow(list(flat.keys()))
            wr.writerow(list(flat.values()))
