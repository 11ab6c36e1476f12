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
