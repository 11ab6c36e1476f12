"""The trainer's callable surfaces around the batch step (src/train.py:154-305, 360-494) and
the behaviours the reference gets from torch objects the fused path replaces: train_epoch's
return dict and log line, train's checkpoint files, the three Inf/NaN warnings, GradScaler's
skipped step, torch.optim.Adam's treatment of parameters without a gradient, the
discriminator's eval mode, and the CSV files of the evaluation driver (src/evaluate.py:294-333)."""
import csv
import importlib
import logging
import os

import numpy as np
import pytest
import torch

import goldgen as G
from conftest import close, close_robust, golden, load_pkg
from test_gpu_parity import load_gold, make_batches, train_cfg

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def sf():
    return load_pkg()


@pytest.fixture(scope="module")
def training(sf):
    return importlib.import_module(sf.__name__ + ".training")


@pytest.mark.parametrize("temporal,use_gan,accum,tag", [("Transformer", True, 2, "trgan"), ("LSTM", False, 1, "lstm")])
def test_train_epoch_returns_reference_dict(sf, training, caplog, temporal, use_gan, accum, tag):
    """Trainer.train_epoch over the golden run's four batches: the returned means equal the
    reference's return value, the running-mean log line has its format, the parameters end where
    the reference's do."""
    g = golden("train")
    cfg = train_cfg(temporal, use_gan, accum)
    cfg["system"]["log_interval"] = 2
    tr = training.Trainer(cfg, device=DEV)
    load_gold(tr.model, "trn_g_" + tag)
    tr.model.temporal_module.dropout = 0.0
    if use_gan:
        load_gold(tr.discriminator, "trn_d_" + tag)
    tr.batch_idx = 7          # train_epoch restarts the batch index (src/train.py:176)
    with caplog.at_level(logging.INFO, logger="Trainer"):
        res = tr.train_epoch(make_batches(4, 2, 3, 64))
    assert list(res.keys()) == ["loss", "image_loss", "station_loss"]
    np.testing.assert_allclose([res["loss"], res["image_loss"], res["station_loss"]], g[f"{tag}_epoch"], rtol=5e-4)
    lines = [r.getMessage() for r in caplog.records if r.name == "Trainer"]
    ref = g[f"{tag}_losses"]
    want = f"Epoch 0 Batch 2: Loss={ref[:2, 0].mean():.6f}"
    assert any(l.startswith(want[:-3]) and ", Img=" in l and ", Stn=" in l for l in lines), lines
    if use_gan:      # batch 4 (index 3) is a discriminator batch: the line carries D_Loss
        assert any(l.startswith("Epoch 0 Batch 4:") and f"D_Loss={ref[3, 4]:.6f}"[:-3] in l for l in lines), lines
    gsd = {k: v.cpu() for k, v in tr.model.state_dict().items()}
    keys = [str(k) for k in g[f"{tag}_gkeys"]]
    cs = G.checksum(gsd, keys)
    np.testing.assert_allclose(np.stack([cs[k] for k in keys])[:, 1], g[f"{tag}_gsum"][:, 1], rtol=5e-4)


def _tiny_dataset(sf, days, cfg):
    synth = importlib.import_module(sf.__name__ + ".synth")
    raw, dates, st = synth.make_dataset_inputs(days, channels=3, height=64, width=64, stations=4)
    return sf.ResidentStreamflowDataset.from_arrays(list(raw), dates, st, cfg)


def test_train_writes_the_reference_checkpoint_files(sf, training, tmp_path):
    """Trainer.train: checkpoint_epoch_{e}.pt after every 10th epoch, gan/gan_epoch_{e}.pt after
    every epoch, best_model.pt on a validation improvement -- each the generator's bare state_dict
    that the reference's Trainer(config, model_path) loads (src/train.py:472-490, 49-51)."""
    cfg = train_cfg("Transformer", True, 1)
    cfg["data"]["output_dir"] = str(tmp_path / "out")
    cfg["data"]["region"] = [100, 110, 25, 35]
    cfg["train"]["batch_size"] = 4
    ds = _tiny_dataset(sf, 5 + 8, cfg)
    tr = training.Trainer(cfg, device=DEV)
    tr.train(ds, num_epochs=10, val_dataset=ds)
    out = tmp_path / "out"
    names = sorted(os.listdir(out))
    assert "checkpoint_epoch_9.pt" in names and "best_model.pt" in names
    assert not any(n.startswith("checkpoint_epoch_") and n != "checkpoint_epoch_9.pt" for n in names)
    assert sorted(os.listdir(out / "gan")) == sorted(f"gan_epoch_{e}.pt" for e in range(10))
    assert any(n.startswith("training_") and n.endswith(".log") for n in names)
    ck = torch.load(out / "checkpoint_epoch_9.pt", map_location="cpu", weights_only=True)
    want = tr.model.state_dict()
    assert list(ck.keys()) == list(want.keys())
    assert all(torch.equal(ck[k], want[k].cpu()) for k in ck)
    tr2 = training.Trainer(cfg, device=DEV, model_path=str(out / "gan" / "gan_epoch_9.pt"))
    assert all(torch.equal(a, b) for a, b in zip(tr2.model.state_dict().values(), tr.model.state_dict().values()))
    assert tr.epoch == 9 and np.isfinite(tr.best_loss)


class _HostWindows(torch.utils.data.Dataset):
    """A map-style dataset of host tensors, the shape of the reference's StreamflowDataset samples
    (src/dataset.py:300-330): served by DataLoader worker processes, not from HBM."""

    def __init__(self, batches):
        self.samples = []
        for b in batches:
            for i in range(b["images"].shape[0]):
                self.samples.append({"images": b["images"][i].cpu(), "output_image": b["output_image"][i].cpu(),
                                     "stations": b["stations"][i], "station_positions": b["station_positions"][i]})
        self.pids = torch.zeros(len(self.samples), dtype=torch.int64).share_memory_()

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, i):
        self.pids[i] = os.getpid()
        return self.samples[i]


def test_train_feeds_map_style_datasets_through_worker_processes(sf, training, tmp_path):
    """A dataset without a loader() method goes through torch's DataLoader as in src/train.py:374-401:
    train.num_workers worker processes (kept between epochs), page-locked batches.  Two epochs over
    12 host-side samples: every sample was produced by a process other than the trainer's, by the
    same workers in both epochs, and the run equals the same batches fed in process (num_workers 0)
    when nothing is shuffled."""
    cfg = train_cfg("LSTM", False, 1)
    cfg["data"]["output_dir"] = str(tmp_path / "out")
    cfg["train"]["batch_size"] = 4
    cfg["train"]["num_workers"] = 2
    ds = _HostWindows(make_batches(3, 4, 3, 64))
    loaders = importlib.import_module(sf.__name__ + ".training")
    cache = {}
    ld = loaders._epoch_loader(ds, 4, False, 0, 0, 1, 2, True, cache)
    first = [b for b in ld]
    assert len(first) == 3 and first[0]["images"].is_pinned() and tuple(first[0]["images"].shape[:1]) == (4,)
    workers = set(ds.pids.tolist())
    assert os.getpid() not in workers and 1 <= len(workers) <= 2
    ds.pids.zero_()
    assert loaders._epoch_loader(ds, 4, False, 1, 0, 1, 2, True, cache) is ld      # the same loader, its workers alive
    second = [b for b in ld]
    assert set(ds.pids.tolist()) == workers
    assert all(torch.equal(a["images"], b["images"]) for a, b in zip(first, second))
    del ld, cache
    torch.manual_seed(3)
    tr = training.Trainer(cfg, device=DEV)
    load_gold(tr.model, "trn_g_lstm")
    tr.train(ds, num_epochs=2)
    assert tr.epoch == 1 and not tr.saw_nan_or_inf()


def test_nan_checks_warn_and_the_step_is_skipped(sf, training, caplog):
    """An Inf in the input images: the three warnings of src/train.py:193-199,221-222 appear, and
    with mixed precision the optimiser step is skipped as GradScaler.step skips it
    (src/train.py:255-260): parameters, moments and step counts stay as they were; the next clean
    batch trains normally."""
    cfg = train_cfg("Transformer", True, 1)
    cfg["system"]["mixed_precision"] = True
    tr = training.Trainer(cfg, device=DEV)
    tr.model.temporal_module.dropout = 0.0
    batches = make_batches(2, 2, 3, 64)
    bad = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batches[0].items()}
    bad["images"][0, 1, 2, 3, 4] = float("inf")
    g0, d0 = tr.opt_g.flat.clone(), tr.opt_d.flat.clone()
    with caplog.at_level(logging.WARNING, logger="Trainer"):
        tr.train_epoch([bad])
    msgs = [r.getMessage() for r in caplog.records]
    for text in ("NaN or Inf found in input images!", "NaN or Inf found in model predictions!", "NaN or Inf found in loss!"):
        assert text in msgs
    assert torch.equal(tr.opt_g.flat, g0) and torch.equal(tr.opt_d.flat, d0)
    assert float(tr.opt_g.m.abs().sum()) == 0.0 and float(tr.opt_g.v.abs().sum()) == 0.0
    assert (tr.opt_g.t, tr.opt_g.skipped, tr.opt_d.t, tr.opt_d.skipped) == (0, 1, 0, 1)
    # train_epoch's own read at the end of the epoch has already reported the skip: the flag says
    # "since the last report", the counters are cumulative
    rep = tr.nan_report()
    assert rep["skipped_steps"] == 1 and rep["skipped_steps_d"] == 1 and not any(rep["flags"])
    tr.train_step(bad["images"], bad["output_image"], bad["station_positions"], bad["stations"])
    rep = tr.nan_report()
    assert rep["flags"] == [True, True, True, True] and rep["skipped_steps"] == 2
    assert not tr.saw_nan_or_inf()          # nothing new since
    caplog.clear()
    with caplog.at_level(logging.WARNING, logger="Trainer"):
        res = tr.train_epoch([batches[1]])
    assert not [r for r in caplog.records if "NaN or Inf" in r.getMessage()]
    assert np.isfinite(res["loss"]) and tr.opt_g.t == 1 and not torch.equal(tr.opt_g.flat, g0)
    assert torch.isfinite(tr.opt_g.flat).all()


def test_flat_adam_against_torch_adam_with_a_missing_gradient_and_a_skipped_step(training):
    """FlatAdam (sf_adam_guarded, step count on the device) against torch.optim.Adam over three
    steps in which one parameter receives no gradient in step 2 (torch leaves it and its moments
    untouched) and a step in between is skipped by the found_inf flag."""
    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Linear(37, 129), torch.nn.Linear(129, 11), torch.nn.Linear(11, 500)).to(DEV)
    ref = torch.nn.Sequential(torch.nn.Linear(37, 129), torch.nn.Linear(129, 11), torch.nn.Linear(11, 500)).to(DEV)
    ref.load_state_dict(net.state_dict())
    opt = training.FlatAdam(net, lr=1e-2, betas=(0.5, 0.999), big_numel=5000, shadow_big=True)   # the 11x500 weight is "big"
    assert [p.numel() for p in opt.big] == [5500]
    topt = torch.optim.Adam(ref.parameters(), lr=1e-2, betas=(0.5, 0.999))
    found = torch.zeros(1, device=DEV, dtype=torch.int32)
    for step in range(4):
        gs = [torch.randn_like(p) for p in net.parameters()]
        missing = 2 if step == 2 else -1
        for i, (p, q, g) in enumerate(zip(net.parameters(), ref.parameters(), gs)):
            p.grad = None if i == missing else g.clone()
            q.grad = None if i == missing else g.clone()
        opt.gather()
        if step == 1:               # a skipped step: nothing may change, torch's optimiser is not stepped
            before = opt.flat.clone()
            found.fill_(1)
            opt.step(1.0, found)
            found.zero_()
            assert torch.equal(opt.flat, before)
            continue
        opt.step(1.0, found)
        topt.step()
        for p, q in zip(net.parameters(), ref.parameters()):
            torch.testing.assert_close(p.detach(), q.detach(), rtol=2e-6, atol=1e-7)
    assert (opt.t, opt.skipped) == (3, 1)
    big = opt.big[0]
    sh = importlib.import_module(training.__name__.rsplit(".", 1)[0] + ".ops").shadow_if_current(big)
    assert sh is not None and torch.equal(sh, big.detach().to(torch.bfloat16))


@pytest.mark.parametrize("size", [64, 50])
def test_discriminator_eval_mode_f32(sf, size):
    """PatchDiscriminator.eval(): running statistics, no buffer updates (golden: the reference's
    module in eval mode, src/model/gan.py:24-64)."""
    g = golden("discriminator_eval")
    d = sf.PatchDiscriminator(1, 16).to(DEV)
    load_gold(d, "disc")
    d.eval()
    before = {k: v.clone() for k, v in d.state_dict().items() if "running" in k or "num_batches" in k}
    x = G.gold_tensor(f"disc/x{size}", (2, 1, size, size)).to(DEV).requires_grad_(True)
    y = d(x)
    close(y, g[f"y{size}"], 2e-5)
    (y * G.gold_tensor(f"disc/gy{size}", tuple(y.shape)).to(DEV)).sum().backward()
    close_robust(x.grad, g[f"gx{size}"], 2e-4)
    close(d.model[3][1].weight.grad, g[f"g{size}_bn_w"], 1e-3)
    close(d.model[3][0].weight.grad, g[f"g{size}_conv_w"], 1e-3)
    after = d.state_dict()
    assert all(torch.equal(after[k], v) for k, v in before.items())
    assert d.last_record is None


def test_evaluation_csv_files_match_the_reference(sf, tmp_path):
    """stations_eval.csv and evaluation_metrics.csv as Evaluator.evaluate_dataset writes them
    (src/evaluate.py:294-333): same file names, column names and order, one row per (date,
    station) in date order, values to 1e-4 (golden: the files of the reference's run)."""
    g, gc = golden("evaluate"), golden("evaluate_csv")
    cfg = {"data": {"upscale_factor": 2, "image_size": [int(g["images0"].shape[-2]), int(g["images0"].shape[-1])],
                    "window_size": int(g["images0"].shape[0])},
           "model": {"input_channels": int(g["images0"].shape[1]), "hidden_dim": 16, "num_layers": 2,
                     "temporal_module": "LSTM"},
           "stations": {"kernel_size": 3}}
    m = sf.StreamflowPredictionModel(cfg).to(DEV)
    load_gold(m, "eval_g")
    ds = [{"images": torch.from_numpy(g[f"images{i}"]), "output_image": torch.from_numpy(g[f"target{i}"]),
           "stations": torch.from_numpy(g[f"stations{i}"]), "station_positions": torch.from_numpy(g[f"pos{i}"]),
           "date": str(gc["dates"][i])} for i in range(int(g["n"]))]
    out = tmp_path / "evalout"
    sf.Evaluator(cfg, model=m).evaluate_dataset(ds, output_dir=str(out), batch_size=3)
    for name in ("stations_eval", "evaluation_metrics"):
        assert name + ".csv" in [str(f) for f in gc["files"]]
        with open(out / (name + ".csv"), newline="") as f:
            rows = list(csv.reader(f))
        assert rows[0] == [str(c) for c in gc[name + "_header"]]
        ref = gc[name + "_rows"]
        assert len(rows) - 1 == ref.shape[0]
        for got, want in zip(rows[1:], ref):
            for col, (a, b) in enumerate(zip(got, want)):
                if name == "stations_eval" and col < 3:
                    assert str(a) == str(b)           # date, position_x, position_y: exact
                else:
                    assert abs(float(a) - float(b)) <= 1e-4 * max(1.0, abs(float(b))), (name, col, a, b)
