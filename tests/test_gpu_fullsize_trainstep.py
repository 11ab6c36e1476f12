"""The whole adversarial train step at the benchmark's model size (10 channels, 200 x 200 ->
400 x 400, Transformer, hidden 128, PatchGAN width 64, 64 stations with 5 x 5 windows;
src/train.py:154-305) in f32 mode, two samples per batch, three batches (a discriminator batch and
two generator-only batches of the reference's schedule): loss trajectory against the CPU oracle's
TrainerOracle on the same weights and inputs (5e-4, the tolerance of the golden trajectory test),
abs-sum checksums of every generator / discriminator tensor after the three fused Adam steps
(1e-3), and the parameter UPDATES themselves, tensor by tensor: norm and direction (cosine) of
(after - before) against the oracle's."""
import importlib

import numpy as np
import pytest
import torch

import goldgen as G
from conftest import NOISE_DRIVEN, PARTLY_NOISE_DRIVEN, load_pkg
from oracle import streamflow_oracle as O
from test_gpu_parity import load_gold

pytestmark = pytest.mark.gpu
CFG = {
    "data": {"image_size": [200, 200], "window_size": 5, "upscale_factor": 2},
    "model": {"input_channels": 10, "hidden_dim": 128, "num_layers": 2, "temporal_module": "Transformer",
              "use_gan": True, "gan_weight": 0.1},
    "train": {"lr": 1e-4},
    "loss": {"metric": "MSE", "weights": {"image": 1.0, "station": 0.5}},
    "stations": {"kernel_size": 5, "metric": "MSE"},
    "system": {"grad_accum_steps": 1, "mixed_precision": False},
}


def _batches(nb, B=2, S=64):
    out = []
    for i in range(nb):
        rs = np.random.RandomState(500 + i)
        pos = [torch.from_numpy(rs.randint(0, 200, size=(S, 2)).astype(np.int64)) for _ in range(B)]
        run = [torch.from_numpy(rs.standard_normal(S).astype(np.float32)) for _ in range(B)]
        out.append({"images": G.gold_tensor(f"fs/img{i}", (B, 5, 10, 200, 200)),
                    "output_image": G.gold_tensor(f"fs/out{i}", (B, 1, 200, 200)),
                    "station_positions": pos, "stations": run})
    return out


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_full_size_train_steps_against_oracle(mode):
    """bf16 (the benchmark's mode: autocast on, bf16 tensors, f32 accumulation) against the same f32
    oracle at the stated bf16 tolerance, 3e-2 of each loss value."""
    import copy
    sf = load_pkg()
    training = importlib.import_module(sf.__name__ + ".training")
    cfg = copy.deepcopy(CFG)
    cfg["system"]["mixed_precision"] = mode == "bf16"
    tol = 5e-4 if mode == "f32" else 3e-2
    tr = training.Trainer(cfg, device="cuda")
    g_sd = load_gold(tr.model, "fs_g")
    d_sd = load_gold(tr.discriminator, "fs_d")
    tr.model.temporal_module.dropout = 0.0
    ref = O.TrainerOracle(CFG, {k: v.clone() for k, v in g_sd.items()}, {k: v.clone() for k, v in d_sd.items()})
    init = {id(tr.model): {k: v.clone() for k, v in g_sd.items()}, id(tr.discriminator): {k: v.clone() for k, v in d_sd.items()}}
    keys = ("total", "image", "station", "gan_g", "gan_d")
    for i, b in enumerate(_batches(3)):
        args = (b["images"], b["output_image"], b["station_positions"], b["stations"])
        o = tr.train_step(*args)
        w = ref.step(*args)
        assert set(k for k in keys if k in o) == set(k for k in keys if k in w)
        # the first batch precedes every parameter update (forward + losses only): in f32 mode it is
        # held to the forward tolerance, 1e-5
        tol_i = 1e-5 if (i == 0 and mode == "f32") else tol
        for k in keys:
            if k in w:
                assert abs(float(o[k].detach() if torch.is_tensor(o[k]) else o[k]) - w[k]) <= tol_i * max(abs(w[k]), 1e-3), (i, k, o[k], w[k])
    assert not tr.saw_nan_or_inf()
    for mod, sd in ((tr.model, ref.g_sd), (tr.discriminator, ref.d_sd)):
        for k, v in mod.state_dict().items():
            if not v.dtype.is_floating_point:
                assert int(v) == int(sd[k]), k
                continue
            a, b_ = float(v.double().abs().sum()), float(sd[k].detach().double().abs().sum())
            assert abs(a - b_) <= 1e-3 * max(b_, 1e-6), (k, a, b_)
    # The updates: three Adam steps of the generator, one of the discriminator.  Adam normalises
    # every entry's step to about lr, so an entry whose gradient is within rounding of zero moves
    # in the direction of the rounding noise: two correct f32 implementations differ in the sign of
    # a fraction f of the entries and reach cos = 1 - 2 f, not 1.  Bounds: f32 cos >= 0.99 and
    # norms to 5e-3; bf16 (a precision statement, the oracle stays f32) cos >= 0.8, norms to 10 %.
    cos_min, norm_tol = (0.99, 5e-3) if mode == "f32" else (0.8, 0.1)
    bad, worst = [], (2.0, "")
    for mod, sd in ((tr.model, ref.g_sd), (tr.discriminator, ref.d_sd)):
        for k, v in mod.state_dict().items():
            if not v.dtype.is_floating_point or "running" in k or k.endswith("pos_encoder.pe") or k in NOISE_DRIVEN:
                continue
            w0 = init[id(mod)][k].double()
            da, db = v.detach().cpu().double() - w0, sd[k].detach().double() - w0
            na, nb = float(da.norm()), float(db.norm())
            assert nb > 0.0, k
            cos = float((da * db).sum()) / max(na * nb, 1e-300)
            need = 0.6 if k.endswith(PARTLY_NOISE_DRIVEN) else cos_min
            if cos < worst[0]:
                worst = (cos, k)
            if cos < need or abs(na - nb) > norm_tol * nb:
                bad.append(f"{k}: cos {cos:.5f} (need {need}), |update| {na:.4e} against {nb:.4e}")
    print(f"[{mode}] least aligned update: {worst[1]} cos {worst[0]:.5f}")
    assert not bad, "\n".join(bad)
