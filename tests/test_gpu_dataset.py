"""The device-resident dataset against the reference's StreamflowDataset (golden fixture
dataset.npz: statistics and every sample of two small directories with masks, NaNs, a resize,
z-score and min-max normalisation, ragged / duplicated / moving station rows) and against the
CPU restatement.  Stated tolerances: statistics 1e-6 relative (float64 sums here, float32
pairwise per-date sums in the reference), samples 1e-5 of the tensor's max, positions exact."""
import numpy as np
import pandas as pd
import pytest
import torch

import goldgen as G
from conftest import close, golden, load_pkg
from oracle import streamflow_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sf():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return load_pkg()


def _make(sf, tag, tmp_path, **kw):
    g = golden("dataset")
    frames, masks, dates, stations, cfg = G.dataset_case(g, tag)
    root = tmp_path / f"case{tag}"
    G.write_dataset_case(str(root), frames, masks, dates, stations)
    cfg["data"]["output_dir"] = str(root / "output")
    ds = sf.ResidentStreamflowDataset(str(root / "images"), str(root / "stations"), cfg, **kw)
    return g, ds, (frames, masks, dates, stations, cfg)


def _by_name(names_have, names_want, *arrays):
    perm = [list(names_have).index(n) for n in names_want]
    return [a[perm] for a in arrays]


@pytest.mark.parametrize("tag", ["A", "B"])
def test_resident_dataset_against_reference(sf, tag, tmp_path):
    g, ds, (frames, masks, dates, stations, cfg) = _make(sf, tag, tmp_path)
    for k in ("image_mean", "image_std", "image_min", "image_max"):
        np.testing.assert_allclose(getattr(ds, k), g[f"{tag}_{k}"], rtol=1e-6)
    np.testing.assert_allclose([ds.runoff_mean, ds.runoff_std, ds.runoff_min, ds.runoff_max],
                               g[f"{tag}_runoff_stats"], rtol=1e-12)
    assert len(ds) == int(g[f"{tag}_n"])
    # the CPU restatement, fed the reference's statistics, for a second opinion on every sample
    fr, mr = O.dataset_prepare(frames, masks, cfg["data"]["image_size"])
    st = {k: g[f"{tag}_{k}"] for k in ("image_mean", "image_std", "image_min", "image_max")}
    st["runoff"] = g[f"{tag}_runoff_stats"]
    for i in range(len(ds)):
        s = ds[i]
        assert s["date"] == str(g[f"{tag}_date{i}"])
        assert s["images"].is_cuda and s["images"].dtype == torch.float32 and s["stations"].dtype == torch.float32
        assert s["station_positions"].dtype == torch.int64
        ref_t = g[f"{tag}_target{i}"]
        got_t = s["output_image"].cpu().numpy()
        assert np.array_equal(np.isnan(got_t), np.isnan(ref_t))          # NaN kept in targets only
        assert np.isfinite(s["images"].cpu().numpy()).all()
        close(s["images"], g[f"{tag}_images{i}"], 1e-5)
        close(np.nan_to_num(got_t), np.nan_to_num(ref_t), 1e-5)
        want_names = [str(n) for n in g[f"{tag}_names{i}"]]
        have_names = ds.sample_station_names(i)
        assert sorted(have_names) == sorted(want_names)
        pos, run = _by_name(have_names, want_names, s["station_positions"].numpy(), s["stations"].numpy())
        assert np.array_equal(pos, g[f"{tag}_pos{i}"])
        np.testing.assert_allclose(run, g[f"{tag}_stations{i}"], rtol=1e-6)
        o = O.dataset_sample(i, fr, mr, dates, stations, st, cfg, have_names)
        close(s["images"], o["images"], 1e-5)
        assert np.array_equal(s["station_positions"].numpy(), o["station_positions"])
    # the reference's cache file was written and a second construction reads it back
    cache = np.load(ds._stats_cache_path(), allow_pickle=False)
    assert [str(x) for x in cache["image_files"]] == [str(p) for p in ds.image_files]
    np.testing.assert_array_equal(cache["image_mean"], ds.image_mean)
    ds2 = sf.ResidentStreamflowDataset(str(ds.image_dir), str(ds.station_dir), cfg)
    assert torch.equal(ds2.frames, ds.frames)


def test_batches_windows_and_packed_stations(sf, tmp_path):
    """batch() equals the collated samples; the packed station rows give the same loss as the
    reference's list-of-tensors form; from_arrays equals the directory constructor."""
    g, ds, (frames, masks, dates, stations, cfg) = _make(sf, "B", tmp_path, use_stats_cache=False)
    idx = [3, 0, 4]
    b = ds.batch(idx)
    samples = [ds[i] for i in idx]
    assert torch.equal(b["images"], torch.stack([s["images"] for s in samples]))
    assert torch.equal(b["output_image"], torch.stack([s["output_image"] for s in samples]))
    assert b["dates"] == [s["date"] for s in samples]
    packed = b["stations"]
    assert packed is b["station_positions"] and packed.S == sum(len(s["stations"]) for s in samples)
    assert packed.offs.cpu().tolist() == np.concatenate([[0], np.cumsum([len(s["stations"]) for s in samples])]).tolist()
    assert torch.equal(packed.pos.cpu().long(), torch.cat([s["station_positions"] for s in samples]))
    loss = sf.CombinedLoss({"loss": {"weights": {"image": 1.0, "station": 0.5}}, "stations": {"kernel_size": 3},
                            "data": {"upscale_factor": 1}})
    torch.manual_seed(0)
    pred = torch.randn(3, 1, 20, 24, device="cuda")
    l_packed = loss(pred, b["output_image"], b["station_positions"], b["stations"])
    l_lists = loss(pred, b["output_image"], [s["station_positions"] for s in samples], [s["stations"] for s in samples])
    for k in ("total", "image", "station"):
        assert float(l_packed[k]) == float(l_lists[k])
    # odd plane size (scalar gather path) through from_arrays, no stations, no normalisation
    rs = np.random.RandomState(0)
    raw = rs.standard_normal((7, 3, 5, 7)).astype(np.float32)
    cfg2 = {"data": {"window_size": 2, "image_size": [5, 7], "flow_threshold": 0.1}}
    d2 = sf.ResidentStreamflowDataset.from_arrays(list(raw), [f"d{i}" for i in range(7)], {}, cfg2, normalize=False)
    want = raw.copy()
    want[:, 0][np.abs(want[:, 0]) < np.float32(0.1)] = 0.0
    bb = d2.batch([4, 1])
    assert np.array_equal(bb["images"].cpu().numpy(), np.stack([want[4:6], want[1:3]]))
    assert np.array_equal(bb["output_image"].cpu().numpy(), np.stack([want[6, 0:1], want[3, 0:1]]))
    assert bb["stations"].S == 0
    # the directory constructor and from_arrays agree
    dfs = {n: pd.DataFrame({"timestamp": [int(t) for t in ts], "lon": lon, "lat": lat, "runoff": run})
           for n, (ts, lon, lat, run) in stations.items()}
    d3 = sf.ResidentStreamflowDataset.from_arrays(list(frames), dates, {n: dfs[n] for n in ds.station_names}, cfg, masks=masks)
    assert torch.equal(d3.frames, ds.frames) and np.array_equal(d3.station_pos, ds.station_pos)
    assert np.array_equal(d3.station_runoff, ds.station_runoff)


def test_loader_feeds_a_train_step(sf, tmp_path):
    """One epoch of sharded batches through Trainer.train_step (packed stations, device batches)."""
    import importlib
    training = importlib.import_module(sf.__name__ + ".training")
    rs = np.random.RandomState(1)
    D, C, S = 14, 3, 64
    raw = rs.gamma(0.7, 3.0, size=(D, C, S, S)).astype(np.float32)
    dates = [f"202002{d + 1:02d}" for d in range(D)]
    st = {f"s{k}": pd.DataFrame({"timestamp": [int(t) for t in dates], "lon": 100.5 + 3 * k, "lat": 33.0 - 2 * k,
                                 "runoff": rs.gamma(2.0, 20.0, size=D)}) for k in range(3)}
    cfg = {"data": {"image_size": [S, S], "window_size": 5, "upscale_factor": 2, "region": [100, 110, 25, 35]},
           "model": {"input_channels": C, "hidden_dim": 16, "num_layers": 2, "temporal_module": "Transformer",
                     "use_gan": True, "gan_weight": 0.1},
           "train": {"lr": 1e-3}, "loss": {"metric": "MSE", "weights": {"image": 1.0, "station": 0.5}},
           "stations": {"kernel_size": 5, "metric": "MSE"}, "system": {"grad_accum_steps": 1, "mixed_precision": False}}
    ds = sf.ResidentStreamflowDataset.from_arrays(list(raw), dates, st, cfg)
    shards = [sf.window_shards(len(ds), 2, True, 7, 0, r, 2, False) for r in range(2)]
    flat = [np.concatenate(s) for s in shards]
    assert len(flat[0]) == len(flat[1]) == len(ds) // 2 and not set(flat[0]) & set(flat[1])
    tr = training.Trainer(cfg, device="cuda")
    losses = []
    for b in ds.loader(2, shuffle=True, seed=7, rank=0, world=2):
        assert b["images"].shape[1:] == (5, C, S, S) and b["output_image"].shape[1:] == (1, S, S)
        o = tr.train_step(b["images"], b["output_image"], b["station_positions"], b["stations"])
        losses.append(float(o["total"].detach()))
        assert float(o["station"]) > 0
    assert len(losses) == len(shards[0]) and np.isfinite(losses).all() and not tr.saw_nan_or_inf()


def test_resident_evaluation_against_reference_metrics(sf):
    """Evaluator.evaluate_dataset over a ResidentStreamflowDataset (frames shared between adjacent
    dates encoded once, image metrics from sf_date_image_metrics on the device, contiguous date
    blocks per rank) against the metrics of the reference's own Evaluator (evaluate.npz) and
    against the host SciPy / NumPy path on the same samples.  Stated tolerance 1e-4 relative
    (float64 window sums here, float32 scipy filters in the reference); NSE to 3 decimals."""
    from test_gpu_parity import load_gold
    g = golden("evaluate")
    n, T = int(g["n"]), int(g["images0"].shape[0])
    C, S = int(g["images0"].shape[1]), int(g["images0"].shape[-1])
    region = [90, 150, -15, 25]
    # the fixture holds normalised samples: rebuild the date sequence they were cut from
    frames = [g["images0"][d] for d in range(T)] + [g[f"images{i}"][T - 1] for i in range(1, n)]
    last = np.zeros((C, S, S), np.float32)
    last[0] = g[f"target{n - 1}"][0]
    frames.append(last)
    for i in range(n):                                    # targets are channel 0 of the next date
        assert np.array_equal(frames[i + T][0], g[f"target{i}"][0])
    dates = [f"202001{d + 1:02d}" for d in range(n + T)]
    ns = int(g["pos0"].shape[0])
    st = {}
    for k in range(ns):
        px, py = float(g["pos0"][k, 0]), float(g["pos0"][k, 1])
        run = [0.0] * T + [float(g[f"stations{i}"][k]) for i in range(n)]
        st[f"s{k}"] = pd.DataFrame({"timestamp": [int(d) for d in dates],
                                    "lon": region[0] + px / (S - 1) * (region[1] - region[0]),
                                    "lat": region[3] - py / (S - 1) * (region[3] - region[2]), "runoff": run})
    cfg = {"data": {"upscale_factor": 2, "image_size": [S, S], "window_size": T, "region": region, "flow_threshold": 0.0},
           "model": {"input_channels": C, "hidden_dim": 16, "num_layers": 2, "temporal_module": "LSTM"},
           "stations": {"kernel_size": 3}}
    ds = sf.ResidentStreamflowDataset.from_arrays(frames, dates, st, cfg, normalize=False)
    for i in range(n):
        assert np.array_equal(ds[i]["images"].cpu().numpy(), g[f"images{i}"])
        assert np.array_equal(ds[i]["station_positions"].numpy(), g[f"pos{i}"])
    m = sf.StreamflowPredictionModel(cfg).to("cuda")
    load_gold(m, "eval_g")
    ev = sf.Evaluator(cfg, model=m)
    keys, skeys = ("RMSE", "MAE", "SSIM", "PSNR", "Correlation"), ("RMSE", "MAE", "R2", "NSE", "MAPE", "KGE")
    host = ev.evaluate_dataset([ds[i] for i in range(n)], batch_size=2)      # list dataset: host metrics path
    for bs in (1, 3, 16):
        res = ev.evaluate_dataset(ds, batch_size=bs)
        np.testing.assert_allclose([res[k] for k in keys], g["metrics"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose([res["station_metrics"][k] for k in skeys], g["station_metrics"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose([res[k] for k in keys], [host[k] for k in keys], rtol=1e-4, atol=1e-5)
    parts = [ev.evaluate_dataset(ds, batch_size=2, shard=(r, 2))["partial"] for r in (0, 1)]
    assert [len(p["per_date"]) for p in parts] == [n // 2, n - n // 2]
    res = sf.merge_partials(parts)
    np.testing.assert_allclose([res[k] for k in keys], g["metrics"], rtol=1e-4, atol=1e-5)
    assert round(res["station_metrics"]["NSE"], 3) == round(float(g["station_metrics"][3]), 3)


def test_date_image_metrics_kernel_against_scipy(sf):
    """sf_date_image_metrics on odd-sized maps (partial tiles, maps narrower than the window's
    halo, a constant pair) against scipy's uniform_filter SSIM and NumPy moments."""
    import importlib
    E = importlib.import_module(sf.__name__ + ".evaluation")
    rs = np.random.RandomState(4)
    for (B, H, W) in [(3, 400, 400), (2, 37, 70), (2, 3, 9), (1, 1, 1), (32, 400, 400)]:   # last: a full evaluation batch
        p = rs.standard_normal((B, H, W)).astype(np.float32)
        t = (0.6 * p + 0.4 * rs.standard_normal((B, H, W))).astype(np.float32)
        if B > 1:
            p[1], t[1] = 2.5, 2.5                                     # constant and equal: zero data range
        out = sf.ops.date_image_metrics(torch.from_numpy(p).cuda(), torch.from_numpy(t).cuda()).cpu().numpy()
        for b in range(B):
            pd64, td64 = p[b].astype(np.float64), t[b].astype(np.float64)
            want = [((pd64 - td64) ** 2).sum(), np.abs(pd64 - td64).sum(), pd64.sum(), td64.sum(), (pd64 ** 2).sum(),
                    (td64 ** 2).sum(), (pd64 * td64).sum(), p[b].min(), p[b].max(), t[b].min(), t[b].max()]
            np.testing.assert_allclose(out[b, :11], want, rtol=1e-12, atol=1e-12)
            i, ssim, psnr, corr = E.per_date_from_moments(7, out[b], H * W)
            assert i == 7
            assert abs(ssim - E.calculate_ssim(p[b], t[b])) < 1e-4
            ref_psnr = E.calculate_psnr(p[b], t[b])
            assert (psnr is None and np.isinf(ref_psnr)) or abs(psnr - ref_psnr) < 1e-3
            if p[b].std() > 1e-10 and t[b].std() > 1e-10:
                assert abs(corr - np.corrcoef(p[b].reshape(-1), t[b].reshape(-1))[0, 1]) < 1e-6
            else:
                assert corr is None
