"""Whole train steps at the workloads the benchmark and the configuration list name, against the
f32 CPU oracle (TrainerOracle restates src/train.py:154-305):
  * the benchmark's own step: 32 samples per batch, bf16, 10 channels 200 x 200 -> 400 x 400,
    Transformer, PatchGAN, 64 stations with 5 x 5 windows -- one full adversarial step (batch 0
    is a discriminator batch), every loss at the stated bf16 tolerance, 3e-2;
  * 128 x 128, Transformer, bf16, batch 32, no GAN: two steps, same tolerance;
  * bf16 against f32 training over 40 steps of the adversarial loop on synthetic data: the loss
    curves stay within 10 % of each other (of the f32 value, per step), the station term -- a mean
    over a few 5 x 5 windows, an order of magnitude smaller than the rest -- within 20 %."""
import copy
import importlib

import numpy as np
import pytest
import torch

import goldgen as G
from conftest import load_pkg
from oracle import streamflow_oracle as O
from test_gpu_parity import load_gold

pytestmark = pytest.mark.gpu


def _cfg(size, use_gan, amp):
    return {
        "data": {"image_size": [size, size], "window_size": 5, "upscale_factor": 2},
        "model": {"input_channels": 10, "hidden_dim": 128, "num_layers": 2, "temporal_module": "Transformer",
                  "use_gan": use_gan, "gan_weight": 0.1},
        "train": {"lr": 1e-4},
        "loss": {"metric": "MSE", "weights": {"image": 1.0, "station": 0.5}},
        "stations": {"kernel_size": 5, "metric": "MSE"},
        "system": {"grad_accum_steps": 1, "mixed_precision": amp},
    }


def _batch(i, B, size, S=64):
    rs = np.random.RandomState(900 + i)
    pos = [torch.from_numpy(rs.randint(0, size, size=(S, 2)).astype(np.int64)) for _ in range(B)]
    run = [torch.from_numpy(rs.standard_normal(S).astype(np.float32)) for _ in range(B)]
    return (G.gold_tensor(f"cfgs/img{size}_{i}", (B, 5, 10, size, size)), G.gold_tensor(f"cfgs/out{size}_{i}", (B, 1, size, size)),
            pos, run)


def _compare(o, w, tol):
    keys = ("total", "image", "station", "gan_g", "gan_d")
    assert set(k for k in keys if k in o) == set(k for k in keys if k in w)
    for k in keys:
        if k in w:
            got = float(o[k].detach())
            assert abs(got - w[k]) <= tol * max(abs(w[k]), 1e-3), (k, got, w[k])


def test_benchmark_step_batch_32_bf16_gan_against_oracle():
    sf = load_pkg()
    training = importlib.import_module(sf.__name__ + ".training")
    torch.set_num_threads(16)
    tr = training.Trainer(_cfg(200, True, True), device="cuda")
    g_sd = load_gold(tr.model, "fs_g")
    d_sd = load_gold(tr.discriminator, "fs_d")
    tr.model.temporal_module.dropout = 0.0
    ref = O.TrainerOracle(_cfg(200, True, False), {k: v.clone() for k, v in g_sd.items()}, {k: v.clone() for k, v in d_sd.items()})
    args = _batch(0, 32, 200)
    o = tr.train_step(*args)
    w = ref.step(*args)
    assert "gan_d" in w
    _compare(o, w, 3e-2)
    assert not tr.saw_nan_or_inf() and tr.opt_g.t == 1 and tr.opt_d.t == 1


def test_config_128_transformer_bf16_batch_32_against_oracle():
    sf = load_pkg()
    training = importlib.import_module(sf.__name__ + ".training")
    torch.set_num_threads(16)
    tr = training.Trainer(_cfg(128, False, True), device="cuda")
    g_sd = load_gold(tr.model, "fs_g")          # the generator's parameter shapes do not depend on the image size
    tr.model.temporal_module.dropout = 0.0
    ref = O.TrainerOracle(_cfg(128, False, False), {k: v.clone() for k, v in g_sd.items()})
    for i in range(2):
        args = _batch(i, 32, 128)
        _compare(tr.train_step(*args), ref.step(*args), 3e-2)
    assert not tr.saw_nan_or_inf() and tr.opt_g.t == 2


def test_bf16_training_tracks_f32_training_over_40_steps():
    sf = load_pkg()
    training = importlib.import_module(sf.__name__ + ".training")
    synth = importlib.import_module(sf.__name__ + ".synth")
    base = {
        "data": {"image_size": [64, 64], "window_size": 5, "upscale_factor": 2, "region": [100, 110, 25, 35]},
        "model": {"input_channels": 4, "hidden_dim": 32, "num_layers": 2, "temporal_module": "Transformer",
                  "use_gan": True, "gan_weight": 0.1},
        "train": {"lr": 2e-4},
        "loss": {"metric": "MSE", "weights": {"image": 1.0, "station": 0.5}},
        "stations": {"kernel_size": 5, "metric": "MSE"},
        "system": {"grad_accum_steps": 1, "mixed_precision": False},
    }

    def l2p(lon, lat):
        px, py = sf.ops.lonlat_to_pixel(torch.from_numpy(lon).cuda(), torch.from_numpy(lat).cuda(), base["data"]["region"], 64, 64)
        return px.cpu().numpy(), py.cpu().numpy()

    batches = synth.make_batches(8, 4, channels=4, height=64, width=64, stations=8, seed=3, lonlat_to_pixel=l2p)
    curves = {}
    init = None
    for amp in (False, True):
        cfg = copy.deepcopy(base)
        cfg["system"]["mixed_precision"] = amp
        torch.manual_seed(11)
        tr = training.Trainer(cfg, device="cuda")
        tr.model.temporal_module.dropout = 0.0
        if init is None:
            init = ({k: v.clone() for k, v in tr.model.state_dict().items()},
                    {k: v.clone() for k, v in tr.discriminator.state_dict().items()})
        else:
            tr.model.load_state_dict(init[0])
            tr.discriminator.load_state_dict(init[1])
        rows = []
        for s in range(40):
            b = batches[s % len(batches)]
            o = tr.train_step(b["images"], b["output_image"], b["station_positions"], b["stations"])
            rows.append([float(o["total"].detach()), float(o["image"]), float(o["station"]), float(o["gan_g"].detach())])
        assert not tr.saw_nan_or_inf()
        curves[amp] = np.array(rows)
    f32, bf = curves[False], curves[True]
    dev = np.abs(bf - f32) / np.maximum(np.abs(f32), 1e-3)
    print("max relative deviation per component (total, image, station, gan_g):", dev.max(axis=0))
    assert f32[-8:, 0].mean() < f32[:8, 0].mean()          # the loop does train
    assert dev[:, [0, 1, 3]].max() <= 0.10 and dev[:, 2].max() <= 0.20, dev.max(axis=0)


def test_benchmark_shape_bf16_gradients_against_f32_gradients():
    """A PRECISION statement, not a parity claim: at the shape the headline number is measured on
    (32 samples, 200 x 200 -> 400 x 400, hidden 128, 64 stations, adversarial term through the
    PatchGAN), every parameter gradient of the generator loss computed in bf16 mode against the
    same HIP path in f32 mode (which the f32 parity tests tie to the reference), same weights and
    inputs, relative L2 per tensor.  Bounds: 5e-2 for weight matrices and conv filters (measured: at
    most 0.04), 0.15 for the first encoder conv (measured 0.09: its gradient has passed through
    every bf16-rounded layer of the model and through the max-pool routing, which flips where two
    candidates round to the same bf16 value), 0.1 for
    the 1-D tensors (biases, norm scales: sums of bf16-rounded maps over few elements) and 0.25
    for tensors that are partly noise driven (an attention in_proj_bias: its key third has an
    analytically zero gradient)."""
    sf = load_pkg()
    B = 32
    cfg = _cfg(200, True, True)
    images, target, pos, run = _batch(0, B, 200)
    images, target = images.cuda(), target.cuda()

    def grads(amp):
        m = sf.StreamflowPredictionModel(cfg).to("cuda")
        load_gold(m, "fs_g")
        m.temporal_module.dropout = 0.0
        d = sf.PatchDiscriminator(1, 64).to("cuda")
        load_gold(d, "fs_d")
        m.train()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            pred = m(images)
            ld = sf.CombinedLoss(cfg)(pred, target, pos, run)
            total = ld["total"] + 0.1 * sf.GANLoss().gan_loss(d(pred), torch.ones(1, device="cuda"))
        total.backward()
        out = {"G." + k: p.grad.detach().float().cpu() for k, p in m.named_parameters()}
        out.update({"D." + k: p.grad.detach().float().cpu() for k, p in d.named_parameters()})
        return out, float(total.detach())

    g32, l32 = grads(False)
    g16, l16 = grads(True)
    assert abs(l16 - l32) <= 3e-2 * abs(l32)
    bad, rows = [], []
    for k, a in g32.items():
        if k[2:] in ("model.2.0.bias", "model.3.0.bias", "model.4.0.bias"):     # zero gradients but for rounding noise
            continue
        b = g16[k]
        rel = float((b.double() - a.double()).norm() / max(float(a.double().norm()), 1e-30))
        bound = 0.25 if k.endswith("in_proj_bias") else (0.1 if a.dim() == 1 else 0.15 if k == "G.cnn_encoder.encoder.0.weight" else 5e-2)
        rows.append((rel / bound, rel, k))
        if rel > bound:
            bad.append(f"{k}: relative L2 error {rel:.3e} > {bound}")
    rows.sort(reverse=True)
    print("largest bf16-vs-f32 gradient errors:", [(k, round(r, 4)) for _, r, k in rows[:6]])
    assert not bad, "\n".join(bad)
