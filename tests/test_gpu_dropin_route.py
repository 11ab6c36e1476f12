"""The drop-in route: the HIP modules driven the way the reference's own Trainer.train_epoch
drives its modules (src/train.py:176-268, restated here, nothing imported from the reference):
two torch.optim.Adam optimisers, one torch.amp.GradScaler, torch.amp.autocast('cuda'), losses
back-propagated through autograd, parameters updated by torch.  With mixed_precision off the
trajectory must equal the reference's golden `trgan` run (the trajectory the package's own fused
trainer is held to); with AMP on (autocast selects the bf16 kernels, the scaler scales the loss
and unscales the gradients) it must track the f32 CPU oracle at the stated bf16 tolerance, 3e-2."""
import numpy as np
import pytest
import torch

import goldgen as G
from conftest import close, golden, load_pkg
from oracle import streamflow_oracle as O
from test_gpu_parity import load_gold, make_batches, train_cfg

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


def _reference_style_epoch(sf, cfg, batches, g_tag, d_tag):
    use_amp = cfg["system"]["mixed_precision"]
    accum = cfg["system"]["grad_accum_steps"]
    gan_weight = cfg["model"]["gan_weight"]
    lr = cfg["train"]["lr"]
    model = sf.StreamflowPredictionModel(cfg).to(DEV)
    g_sd = load_gold(model, g_tag)
    model.temporal_module.dropout = 0.0
    loss_fn = sf.CombinedLoss(cfg).to(DEV)
    disc = sf.PatchDiscriminator(input_channels=1, hidden_channels=64).to(DEV)
    d_sd = load_gold(disc, d_tag)
    gan_loss_fn = sf.GANLoss(lambda_gan=gan_weight, lambda_pixel=1.0).to(DEV)
    optimizer_D = torch.optim.Adam(disc.parameters(), lr=lr * 0.2, betas=(0.5, 0.999))
    optimizer = torch.optim.Adam(model.parameters(), lr=lr)
    scaler = torch.amp.GradScaler("cuda") if use_amp else None
    model.train()
    rows = []
    for batch_idx, batch in enumerate(batches):
        images = batch["images"].to(DEV, non_blocking=True)
        output_image = batch["output_image"].to(DEV, non_blocking=True)
        positions = [p.to(DEV) for p in batch["station_positions"]]
        runoffs = [r.to(DEV) for r in batch["stations"]]
        with torch.amp.autocast(device_type="cuda", enabled=use_amp):
            predictions = model(images)
            loss_dict = loss_fn(predictions, output_image, positions, runoffs)
            loss_G = loss_dict["total"]
            pred_fake = disc(predictions)
            loss_G_gan = gan_loss_fn.gan_loss(pred_fake, torch.ones_like(pred_fake))
            loss_G = loss_G + gan_weight * loss_G_gan
            loss_dict["gan_g"] = loss_G_gan
            loss_dict["total"] = loss_G
        if use_amp:
            scaler.scale(loss_G / accum).backward()
        else:
            (loss_G / accum).backward()
        if batch_idx % 3 == 0:
            with torch.amp.autocast(device_type="cuda", enabled=use_amp):
                pred_real = disc(output_image)
                pred_fake_d = disc(predictions.detach())
                loss_D = gan_loss_fn.discriminator_loss(pred_real, pred_fake_d)
                loss_dict["gan_d"] = loss_D
            if use_amp:
                scaler.scale(loss_D / accum).backward()
            else:
                (loss_D / accum).backward()
        rows.append([loss_dict["total"].item(), loss_dict["image"].item(), loss_dict["station"].item(),
                     loss_dict["gan_g"].item(), loss_dict["gan_d"].item() if "gan_d" in loss_dict else np.nan])
        if (batch_idx + 1) % accum == 0:
            if use_amp:
                scaler.step(optimizer)
                if batch_idx % 3 == 0:
                    scaler.step(optimizer_D)
                scaler.update()
            else:
                optimizer.step()
                if batch_idx % 3 == 0:
                    optimizer_D.step()
            optimizer.zero_grad()
            optimizer_D.zero_grad()
    return np.array(rows), model, disc, g_sd, d_sd


def test_reference_loop_over_hip_modules_f32_matches_the_golden_trajectory():
    sf = load_pkg()
    g = golden("train")
    cfg = train_cfg("Transformer", True, 2)
    rows, model, disc, _, _ = _reference_style_epoch(sf, cfg, make_batches(4, 2, 3, 64), "trn_g_trgan", "trn_d_trgan")
    ref = g["trgan_losses"]
    assert np.array_equal(np.isnan(rows), np.isnan(ref))
    np.testing.assert_allclose(rows[0], ref[0], rtol=1e-5)               # before any update: forward + loss only
    np.testing.assert_allclose(np.nan_to_num(rows), np.nan_to_num(ref), rtol=5e-4)
    gsd = {k: v.cpu() for k, v in model.state_dict().items()}
    keys = [str(k) for k in g["trgan_gkeys"]]
    cs = G.checksum(gsd, keys)
    np.testing.assert_allclose(np.stack([cs[k] for k in keys])[:, 1], g["trgan_gsum"][:, 1], rtol=5e-4)
    dsd = {k: v.cpu() for k, v in disc.state_dict().items()}
    keys = [str(k) for k in g["trgan_dkeys"]]
    cs = G.checksum(dsd, keys)
    np.testing.assert_allclose(np.stack([cs[k] for k in keys])[:, 1], g["trgan_dsum"][:, 1], rtol=5e-4)
    close(dsd["model.2.1.running_var"], g["trgan_d_bn2_rv"], 1e-4)


def test_reference_loop_over_hip_modules_with_autocast_and_gradscaler():
    sf = load_pkg()
    cfg = train_cfg("Transformer", True, 2)
    cfg["system"]["mixed_precision"] = True
    batches = make_batches(4, 2, 3, 64)
    rows, model, disc, g_sd, d_sd = _reference_style_epoch(sf, cfg, batches, "trn_g_trgan", "trn_d_trgan")
    ocfg = train_cfg("Transformer", True, 2)
    ref = O.TrainerOracle(ocfg, {k: v.clone() for k, v in g_sd.items()}, {k: v.clone() for k, v in d_sd.items()})
    for i, b in enumerate(batches):
        w = ref.step(b["images"], b["output_image"], b["station_positions"], b["stations"])
        for col, k in enumerate(("total", "image", "station", "gan_g", "gan_d")):
            if k in w:
                assert abs(rows[i, col] - w[k]) <= 3e-2 * max(abs(w[k]), 1e-3), (i, k, rows[i, col], w[k])
            else:
                assert np.isnan(rows[i, col])
    # the scaler did not skip a step: every parameter moved, none is non-finite
    for n, p in list(model.named_parameters()) + list(disc.named_parameters()):
        assert torch.isfinite(p).all(), n
    moved = [float((p.detach().cpu() - g_sd[n]).abs().max()) > 0 for n, p in model.named_parameters()]
    assert all(moved)
