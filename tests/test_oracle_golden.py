"""Pins the CPU oracle against vectors produced by the reference itself
(tests/golden/make_golden.py).  Runs on CPU."""
import numpy as np
import pytest
import torch

import goldgen as G
from conftest import golden, close, close_robust, check_update_projection, check_update_signs
from oracle import streamflow_oracle as O

torch.set_num_threads(8)


# shapes of the reference modules' state_dicts, restated from src/model/*.py
def encoder_shapes(cin, hd, prefix=""):
    s = {}
    ch = [cin, hd, hd * 2]
    for li, idx in enumerate((0, 4)):
        s[f"{prefix}encoder.{idx}.weight"] = (ch[li + 1], ch[li], 3, 3)
        for nm in ("weight", "bias", "running_mean", "running_var"):
            s[f"{prefix}encoder.{idx + 1}.{nm}"] = (ch[li + 1],)
        s[f"{prefix}encoder.{idx + 1}.num_batches_tracked"] = ()
    return s


def transformer_shapes(din, hd, prefix="", layers=2):
    s = {f"{prefix}input_proj.weight": (hd, din), f"{prefix}input_proj.bias": (hd,),
         f"{prefix}pos_encoder.pe": (1, 100, hd)}
    for l in range(layers):
        p = f"{prefix}transformer_encoder.layers.{l}."
        s[p + "self_attn.in_proj_weight"] = (3 * hd, hd)
        s[p + "self_attn.in_proj_bias"] = (3 * hd,)
        s[p + "self_attn.out_proj.weight"] = (hd, hd)
        s[p + "self_attn.out_proj.bias"] = (hd,)
        s[p + "linear1.weight"] = (4 * hd, hd)
        s[p + "linear1.bias"] = (4 * hd,)
        s[p + "linear2.weight"] = (hd, 4 * hd)
        s[p + "linear2.bias"] = (hd,)
        for n in ("norm1", "norm2"):
            s[p + n + ".weight"] = (hd,)
            s[p + n + ".bias"] = (hd,)
    return s


def lstm_shapes(din, hd, prefix="", layers=2):
    s = {f"{prefix}input_proj.weight": (hd, din), f"{prefix}input_proj.bias": (hd,)}
    for l in range(layers):
        s[f"{prefix}lstm.weight_ih_l{l}"] = (4 * hd, hd)
        s[f"{prefix}lstm.weight_hh_l{l}"] = (4 * hd, hd)
        s[f"{prefix}lstm.bias_ih_l{l}"] = (4 * hd,)
        s[f"{prefix}lstm.bias_hh_l{l}"] = (4 * hd,)
    return s


def decoder_shapes(cin, prefix="", hc=64):
    return {f"{prefix}conv1.weight": (hc, cin, 3, 3), f"{prefix}conv1.bias": (hc,),
            f"{prefix}conv2.weight": (hc, hc, 3, 3), f"{prefix}conv2.bias": (hc,),
            f"{prefix}output_conv.weight": (1, hc, 1, 1), f"{prefix}output_conv.bias": (1,)}


def disc_shapes(hc=64, cin=1):
    s = {"model.0.weight": (hc, cin, 4, 4), "model.0.bias": (hc,)}
    ch = [hc, hc * 2, hc * 4, hc * 8]
    for i, li in enumerate((2, 3, 4)):
        s[f"model.{li}.0.weight"] = (ch[i + 1], ch[i], 4, 4)
        s[f"model.{li}.0.bias"] = (ch[i + 1],)
        for nm in ("weight", "bias", "running_mean", "running_var"):
            s[f"model.{li}.1.{nm}"] = (ch[i + 1],)
        s[f"model.{li}.1.num_batches_tracked"] = ()
    s["model.5.weight"] = (1, hc * 8, 4, 4)
    s["model.5.bias"] = (1,)
    return s


def generator_shapes(cin, hd, temporal):
    s = encoder_shapes(cin, hd, "cnn_encoder.")
    feat = hd * 2 * 32 * 32
    if temporal.upper() == "TRANSFORMER":
        s.update(transformer_shapes(feat, hd, "temporal_module."))
    else:
        s.update(lstm_shapes(feat, hd, "temporal_module."))
    s["temporal_to_spatial.weight"] = (feat, hd)
    s["temporal_to_spatial.bias"] = (feat,)
    s.update(decoder_shapes(hd * 2, "decoder."))
    return s


def gstate(shapes, tag):
    keep = {k: O.positional_encoding(v[1], v[2]) for k, v in shapes.items() if k.endswith("pos_encoder.pe")}
    return G.gold_state(shapes, tag, keep=keep)


# ------------------------------------------------------------------ tests
def test_pixel_map_bit_exact():
    g = golden("pixel_map")
    for ci, (region, w, h) in enumerate(G.PIXEL_CASES):
        lon, lat = G.pixel_sweep(region, w, h)
        px, py = O.lonlat_to_pixel(lon, lat, region, w, h)
        assert np.array_equal(px, g[f"px{ci}"].astype(np.int64))
        assert np.array_equal(py, g[f"py{ci}"].astype(np.int64))


@pytest.mark.parametrize("size", [64, 128, 200])
def test_encoder_train(size):
    g = golden("encoder")
    sd = gstate(encoder_shapes(4, 16), "enc")
    x = G.gold_tensor(f"enc/x{size}", (2, 4, size, size))
    y, upd = O.encoder_forward(x, sd, prefix="", training=True)
    close(y, g[f"train{size}"])
    close(upd["encoder.1.running_mean"], g[f"rm1_{size}"])
    close(upd["encoder.1.running_var"], g[f"rv1_{size}"])
    close(upd["encoder.5.running_mean"], g[f"rm2_{size}"])
    close(upd["encoder.5.running_var"], g[f"rv2_{size}"])


def test_encoder_eval_and_grads():
    g = golden("encoder")
    sd = gstate(encoder_shapes(4, 16), "enc")
    x = G.gold_tensor("enc/x128", (2, 4, 128, 128))
    y, _ = O.encoder_forward(x, sd, prefix="", training=False)
    close(y, g["eval128"])
    for k in sd:
        if sd[k].dtype.is_floating_point and "running" not in k:
            sd[k].requires_grad_(True)
    x = G.gold_tensor("enc/x200", (2, 4, 200, 200)).requires_grad_(True)
    y, _ = O.encoder_forward(x, sd, prefix="", training=True)
    gy = G.gold_tensor("enc/gy200", tuple(y.shape))
    (y * gy).sum().backward()
    close(sd["encoder.0.weight"].grad, g["g200_w0"], 2e-5)
    close(sd["encoder.4.weight"].grad, g["g200_w4"], 2e-5)
    close(sd["encoder.1.weight"].grad, g["g200_bn1w"], 2e-5)
    close(sd["encoder.1.bias"].grad, g["g200_bn1b"], 2e-5)
    close(sd["encoder.5.weight"].grad, g["g200_bn5w"], 2e-5)
    close(sd["encoder.5.bias"].grad, g["g200_bn5b"], 2e-5)
    close_robust(x.grad[:, :, ::5, ::5], g["g200_x_sub"], 2e-5)


@pytest.mark.parametrize("kind", ["tr", "lstm"])
def test_temporal(kind):
    g = golden("temporal")
    B, T, D_in, Hd = 3, 5, 256, 32
    shapes = transformer_shapes(D_in, Hd) if kind == "tr" else lstm_shapes(D_in, Hd)
    sd = gstate(shapes, "tmp_" + kind)
    fwd = O.transformer_forward if kind == "tr" else O.lstm_forward
    x0 = G.gold_tensor("tmp/x", (B, T, D_in))
    with torch.no_grad():
        close(fwd(x0, sd, prefix=""), g[f"{kind}_y_eval"], 2e-5)
    for k in sd:
        if not k.endswith("pe"):
            sd[k].requires_grad_(True)
    x = x0.clone().requires_grad_(True)
    y = fwd(x, sd, prefix="")
    close(y, g[f"{kind}_y"])
    (y * G.gold_tensor("tmp/gy", (B, Hd))).sum().backward()
    close(x.grad, g[f"{kind}_gx"], 2e-5)
    for k in sd:
        if not k.endswith("pe"):
            close(sd[k].grad, g[f"{kind}_g_{k}"], 5e-5)


@pytest.mark.parametrize("oh", [64, 100])
def test_decoder(oh):
    g = golden("decoder")
    sd = gstate(decoder_shapes(16), "dec")
    for k in sd:
        sd[k].requires_grad_(True)
    x = G.gold_tensor("dec/x", (2, 16, 32, 32)).requires_grad_(True)
    y = O.decoder_forward(x, sd, oh, oh, prefix="")
    close(y, g[f"y{oh}"])
    (y * G.gold_tensor(f"dec/gy{oh}", tuple(y.shape))).sum().backward()
    close_robust(x.grad, g[f"gx{oh}"], 2e-5)
    for k in sd:
        close(sd[k].grad, g[f"g{oh}_{k}"], 2e-5 if oh == 64 else 1e-3)  # a ReLU flip perturbs every weight grad a little


@pytest.mark.parametrize("size", [64, 50])
def test_discriminator(size):
    g = golden("discriminator")
    sd = gstate(disc_shapes(16), "disc")
    for k in sd:
        if sd[k].dtype.is_floating_point and "running" not in k:
            sd[k].requires_grad_(True)
    x = G.gold_tensor(f"disc/x{size}", (2, 1, size, size)).requires_grad_(True)
    y, upd = O.discriminator_forward(x, sd)
    close(y, g[f"y{size}"], 2e-5)
    (y * G.gold_tensor(f"disc/gy{size}", tuple(y.shape))).sum().backward()
    close_robust(x.grad, g[f"gx{size}"], 1e-4)
    for k, v in upd.items():
        if "running" in k:
            close(v, g[f"buf{size}_{k}"])
    for k in sd:
        if sd[k].requires_grad:
            if k in ("model.2.0.bias", "model.3.0.bias", "model.4.0.bias"):
                # a conv bias in front of a train-mode BatchNorm has zero gradient: rounding noise
                assert float(sd[k].grad.abs().max()) < 1e-4 and float(np.abs(g[f"g{size}_{k}"]).max()) < 1e-4
            else:
                close(sd[k].grad, g[f"g{size}_{k}"], 1e-3)


def station_cases(H, W):
    pos = [
        torch.tensor([[0, 0], [W - 1, H - 1], [5, 7], [6, 7], [5, 7], [W - 1, 0], [1, H - 2]], dtype=torch.int64),
        torch.zeros((0, 2), dtype=torch.int64),
        torch.tensor([[10, 3], [11, 4], [W // 2, H // 2]], dtype=torch.int64),
    ]
    run = [G.gold_tensor("loss/r0", (7,)), torch.zeros((0,)), G.gold_tensor("loss/r2", (3,))]
    return pos, run


def loss_inputs():
    B, H, W = 3, 40, 48
    pred0 = G.gold_tensor("loss/pred", (B, 1, H, W))
    tgt = G.gold_tensor("loss/tgt", (B, 1, H // 2, W // 2))
    tgt[:, :, :5, :7] = 0.0
    tgt_same = G.gold_tensor("loss/tgt_same", (B, 1, H, W))
    tgt_same[:, :, 10:20, :] = 0.0
    pos, run = station_cases(H, W)
    return pred0, tgt, tgt_same, pos, run


def test_losses():
    g = golden("losses")
    pred0, tgt, tgt_same, pos, run = loss_inputs()
    for metric in ("MSE", "MAE"):
        for nm, t in (("up", tgt), ("same", tgt_same)):
            p = pred0.clone().requires_grad_(True)
            l = O.image_loss(p, t, metric)
            l.backward()
            close(l, g[f"img_{metric}_{nm}"])
            close(p.grad, g[f"img_{metric}_{nm}_g"])
        for k in (3, 5):
            p = pred0.clone().requires_grad_(True)
            l = O.station_loss(p, pos, run, k, metric)
            l.backward()
            close(l, g[f"stn_{metric}_{k}"])
            close(p.grad, g[f"stn_{metric}_{k}_g"])
    cfg = {"loss": {"metric": "MSE", "weights": {"image": 1.0, "station": 0.5}},
           "stations": {"kernel_size": 5, "metric": "MSE"}}
    p = pred0.clone().requires_grad_(True)
    ld = O.combined_loss(p, tgt, pos, run, cfg)
    ld["total"].backward()
    close(ld["total"], g["comb_total"])
    close(ld["image"], g["comb_image"])
    close(ld["station"], g["comb_station"])
    close(p.grad, g["comb_g"])
    B = pred0.shape[0]
    ld = O.combined_loss(pred0, tgt, [torch.zeros((0, 2), dtype=torch.int64)] * B, [torch.zeros(0)] * B, cfg)
    close(ld["total"], g["comb_nostn_total"])
    assert float(ld["station"]) == float(g["comb_nostn_station"]) == 0.0
    lr = G.gold_tensor("loss/logit_real", (B, 1, 5, 6), 2.0)
    lf = G.gold_tensor("loss/logit_fake", (B, 1, 11, 12), 2.0)
    a = lf.clone().requires_grad_(True)
    l = O.bce_with_logits_mean(a, 1.0)
    l.backward()
    close(l, g["gan_g"])
    close(a.grad, g["gan_g_grad"])
    a = lr.clone().requires_grad_(True)
    b = lf.clone().requires_grad_(True)
    l = O.discriminator_loss(a, b)
    l.backward()
    close(l, g["gan_d"])
    close(a.grad, g["gan_d_greal"])
    close(b.grad, g["gan_d_gfake"])
    pv = G.gold_array("loss/mp", (200,), 1.0, np.float64) + 3.0
    tv = pv + G.gold_array("loss/mt", (200,), 0.5, np.float64)
    m = O.calculate_metrics(pv, tv)
    np.testing.assert_allclose([m[k] for k in ("RMSE", "MAE", "R2", "NSE", "MAPE", "KGE")], g["metrics"],
                               rtol=1e-12)


@pytest.mark.parametrize("temporal", ["LSTM", "Transformer"])
def test_generator_small(temporal):
    g = golden("generator")
    sd = gstate(generator_shapes(3, 16, temporal), "gen_" + temporal)
    x = G.gold_tensor("gen/x", (2, 5, 3, 64, 64))
    with torch.no_grad():
        y, _ = O.generator_forward(x, sd, temporal, training=False)
        close(y, g[f"{temporal}_eval"], 2e-5)
        y, _ = O.generator_forward(x, sd, temporal, training=True)
        close(y, g[f"{temporal}_train"], 2e-5)


def test_generator_full_size():
    g = golden("generator")
    sd = gstate(generator_shapes(10, 128, "Transformer"), "gen_full")
    x = G.gold_tensor("gen_full/x", (1, 5, 10, 200, 200))
    with torch.no_grad():
        y, _ = O.generator_forward(x, sd, "Transformer", training=False)
    close(y, g["full_eval"], 2e-5)


def train_cfg(temporal, use_gan, accum):
    return {
        "data": {"image_size": [64, 64], "window_size": 5, "upscale_factor": 2},
        "model": {"input_channels": 3, "hidden_dim": 16, "num_layers": 2, "temporal_module": temporal,
                  "use_gan": use_gan, "gan_weight": 0.1},
        "train": {"lr": 1e-3},
        "loss": {"metric": "MSE", "weights": {"image": 1.0, "station": 0.5}},
        "stations": {"kernel_size": 5, "metric": "MSE"},
        "system": {"grad_accum_steps": accum},
    }


def make_batches(nb, B, C, size, k_seed="trn"):
    batches = []
    for i in range(nb):
        pos, run = [], []
        for b in range(B):
            n = [3, 0, 4, 2][(i + b) % 4]
            rs = np.random.RandomState(100 * i + b)
            pos.append(torch.from_numpy(rs.randint(0, size, size=(n, 2)).astype(np.int64)))
            run.append(torch.from_numpy(rs.standard_normal(n).astype(np.float32)))
        batches.append({
            "images": G.gold_tensor(f"{k_seed}/img{i}", (B, 5, C, size, size)),
            "output_image": G.gold_tensor(f"{k_seed}/out{i}", (B, 1, size, size)),
            "station_positions": pos, "stations": run,
        })
    return batches


@pytest.mark.parametrize("temporal,use_gan,accum,tag", [("Transformer", True, 2, "trgan"), ("LSTM", False, 1, "lstm")])
def test_train_epoch_semantics(temporal, use_gan, accum, tag):
    """Four batches of Trainer.train_epoch: loss trajectory, the D-schedule quirk and final
    parameter checksums."""
    g = golden("train")
    cfg = train_cfg(temporal, use_gan, accum)
    gsd = gstate(generator_shapes(3, 16, temporal), "trn_g_" + tag)
    dsd = gstate(disc_shapes(64), "trn_d_" + tag) if use_gan else None
    tr = O.TrainerOracle(cfg, gsd, dsd)
    g_init = {k: v.detach().clone() for k, v in tr.g_sd.items()}
    d_init = {k: v.detach().clone() for k, v in tr.d_sd.items()} if use_gan else None
    rows = []
    for b in make_batches(4, 2, 3, 64):
        o = tr.step(b["images"], b["output_image"], b["station_positions"], b["stations"])
        rows.append([o["total"], o["image"], o["station"], o.get("gan_g", np.nan), o.get("gan_d", np.nan)])
    rows = np.array(rows)
    ref = g[f"{tag}_losses"]
    assert np.array_equal(np.isnan(rows), np.isnan(ref))
    np.testing.assert_allclose(np.nan_to_num(rows), np.nan_to_num(ref), rtol=2e-4)
    keys = [str(k) for k in g[f"{tag}_gkeys"]]
    cs = G.checksum(tr.g_sd, keys)
    got = np.stack([cs[k] for k in keys])
    np.testing.assert_allclose(got[:, 1], g[f"{tag}_gsum"][:, 1], rtol=2e-4)
    check_update_projection(tr.g_sd, g_init, keys, g[f"{tag}_gproj"])
    close(tr.g_sd["decoder.output_conv.weight"], g[f"{tag}_g_dec_out_w"], 1e-3)
    close(tr.g_sd["cnn_encoder.encoder.1.running_mean"], g[f"{tag}_g_bn1_rm"], 1e-4)
    if use_gan:
        dkeys = [str(k) for k in g[f"{tag}_dkeys"]]
        # one Adam step in these four batches: every entry moved by lr, so the norms are exact, the
        # signs are the update (a projection would only sum 2 lr r_k over the few noise-level entries that flip)
        check_update_projection(tr.d_sd, d_init, dkeys, g[f"{tag}_dproj"], tol_proj=None)
        check_update_signs(tr.d_sd, d_init, dkeys, g[f"{tag}_dsign"])
        keys = [str(k) for k in g[f"{tag}_dkeys"]]
        cs = G.checksum(tr.d_sd, keys)
        got = np.stack([cs[k] for k in keys])
        np.testing.assert_allclose(got[:, 1], g[f"{tag}_dsum"][:, 1], rtol=2e-4)
        close(tr.d_sd["model.5.bias"], g[f"{tag}_d_m5_b"], 1e-3)
        close(tr.d_sd["model.2.1.running_var"], g[f"{tag}_d_bn2_rv"], 1e-4)


def test_evaluate_path():
    g = golden("evaluate")
    cfg = {"data": {"upscale_factor": 2}, "model": {"temporal_module": "LSTM", "num_layers": 2},
           "stations": {"kernel_size": 3}}
    sd = gstate(generator_shapes(2, 16, "LSTM"), "eval_g")
    samples = []
    for i in range(int(g["n"])):
        samples.append({"images": torch.from_numpy(g[f"images{i}"]), "output_image": torch.from_numpy(g[f"target{i}"]),
                        "stations": torch.from_numpy(g[f"stations{i}"]),
                        "station_positions": torch.from_numpy(g[f"pos{i}"])})
    res = O.evaluate_samples(samples, sd, cfg)
    got = np.array([res[k] for k in ("RMSE", "MAE", "SSIM", "PSNR", "Correlation")])
    np.testing.assert_allclose(got, g["metrics"], rtol=1e-4, atol=1e-5)
    sm = res["station_metrics"]
    np.testing.assert_allclose([sm[k] for k in ("RMSE", "MAE", "R2", "NSE", "MAPE", "KGE")],
                               g["station_metrics"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("tag", ["A", "B"])
def test_dataset_restatement(oracle, tag):
    """Statistics and every sample of StreamflowDataset (masks, NaNs, threshold, resize, minmax
    and z-score, ragged station rows).  Statistics to 1e-6 relative (the reference sums float32
    per date with numpy's pairwise order, so does the restatement); samples to 1e-5 of the max."""
    import goldgen as G
    g = golden("dataset")
    frames, masks, dates, stations, cfg = G.dataset_case(g, tag)
    fr, mr = oracle.dataset_prepare(frames, masks, cfg["data"]["image_size"])
    st = oracle.dataset_statistics(fr, mr, [stations[n][3] for n in G.DATASET_STATIONS])
    for k in ("image_mean", "image_std", "image_min", "image_max"):
        np.testing.assert_allclose(st[k], g[f"{tag}_{k}"], rtol=1e-6)
    np.testing.assert_allclose(st["runoff"], g[f"{tag}_runoff_stats"], rtol=1e-12)
    order = [str(n) for n in g[f"{tag}_order"]]
    assert int(g[f"{tag}_n"]) == len(dates) - cfg["data"]["window_size"]
    for i in range(int(g[f"{tag}_n"])):
        s = oracle.dataset_sample(i, fr, mr, dates, stations, st, cfg, order)
        assert s["date"] == str(g[f"{tag}_date{i}"])
        ref_img, ref_t = g[f"{tag}_images{i}"], g[f"{tag}_target{i}"]
        assert np.array_equal(np.isnan(s["output_image"]), np.isnan(ref_t))
        close(np.nan_to_num(s["images"]), np.nan_to_num(ref_img), 1e-5)
        close(np.nan_to_num(s["output_image"]), np.nan_to_num(ref_t), 1e-5)
        assert np.array_equal(s["station_positions"], g[f"{tag}_pos{i}"])
        np.testing.assert_allclose(s["stations"], g[f"{tag}_stations{i}"], rtol=1e-6)


def baseline_shapes(name, cin=3, hidden=32, layers=2, window=5):
    """state_dict shapes of the baseline models, restated from src/model/baseline_models.py."""
    s = {}
    ch = [cin, 32, 32, 16]
    for li, i in enumerate((0, 3, 6)):
        s[f"reducer.encoder.{i}.weight"] = (ch[li + 1], ch[li], 3, 3)
        for nm in ("weight", "bias", "running_mean", "running_var"):
            s[f"reducer.encoder.{i + 1}.{nm}"] = (ch[li + 1],)
        s[f"reducer.encoder.{i + 1}.num_batches_tracked"] = ()
    if name in ("LSTM", "Attention-GRU"):
        rnn, gates = ("lstm", 4) if name == "LSTM" else ("gru", 3)
        s["input_proj.weight"], s["input_proj.bias"] = (hidden, 1024), (hidden,)
        for l in range(layers):
            s[f"{rnn}.weight_ih_l{l}"], s[f"{rnn}.weight_hh_l{l}"] = (gates * hidden, hidden), (gates * hidden, hidden)
            s[f"{rnn}.bias_ih_l{l}"], s[f"{rnn}.bias_hh_l{l}"] = (gates * hidden,), (gates * hidden,)
        if name == "Attention-GRU":
            for nm in ("query", "key", "value"):
                s[f"attention.{nm}.weight"], s[f"attention.{nm}.bias"] = (hidden, hidden), (hidden,)
        head_in = hidden
    else:
        dims = [window * 1024, 4 * hidden, 4 * hidden, 2 * hidden]
        for j, i in enumerate((0, 3, 6)):
            s[f"mlp.{i}.weight"], s[f"mlp.{i}.bias"] = (dims[j + 1], dims[j]), (dims[j + 1],)
        head_in = 2 * hidden
    s["head_proj.weight"], s["head_proj.bias"] = (4096, head_in), (4096,)
    s["decoder.0.weight"], s["decoder.0.bias"] = (64, 64, 3, 3), (64,)
    s["decoder.2.weight"], s["decoder.2.bias"] = (1, 64, 1, 1), (1,)
    return s


@pytest.mark.parametrize("name", ["LSTM", "BP", "Attention-GRU"])
def test_baseline_models_restatement(name):
    g = golden("baselines")
    shapes = baseline_shapes(name)
    assert list(shapes.keys()) == [str(k) for k in g[f"{name}_keys"]]          # the reference's keys, in its order
    sd = G.gold_state(shapes, "base_" + name)
    x = G.gold_tensor("base/x", (2, 5, 3, 64, 64))
    y, _ = O.baseline_forward(x, sd, name, 128, 128, training=False)
    close(y, g[f"{name}_eval"], 1e-5)
    y, new = O.baseline_forward(x, sd, name, 128, 128, training=True)
    close(y, g[f"{name}_train"], 1e-5)
    close(new["reducer.encoder.7.running_mean"], g[f"{name}_rm7"], 1e-5)
    close(new["reducer.encoder.1.running_var"], g[f"{name}_rv1"], 1e-5)


@pytest.mark.parametrize("size", [64, 50])
def test_discriminator_eval_mode(size):
    """PatchDiscriminator.eval(): BatchNorm on its running statistics, no buffer updates
    (golden from the reference's module in eval mode)."""
    g = golden("discriminator_eval")
    sd = gstate(disc_shapes(16), "disc")
    for k in sd:
        if sd[k].dtype.is_floating_point and "running" not in k:
            sd[k].requires_grad_(True)
    x = G.gold_tensor(f"disc/x{size}", (2, 1, size, size)).requires_grad_(True)
    y, upd = O.discriminator_forward(x, sd, training=False)
    assert not upd
    close(y, g[f"y{size}"], 2e-5)
    (y * G.gold_tensor(f"disc/gy{size}", tuple(y.shape))).sum().backward()
    close_robust(x.grad, g[f"gx{size}"], 1e-4)
    close(sd["model.3.1.weight"].grad, g[f"g{size}_bn_w"], 1e-3)
    close(sd["model.3.0.weight"].grad, g[f"g{size}_conv_w"], 1e-3)


# ------------------------------------------------------------------ constructor options
OPTION_CASES = (("nobn", 3, 1, False, 72), ("k4p2", 4, 2, True, 70), ("k1p0nobn", 1, 0, False, 64), ("k2p0", 2, 0, True, 90))


def encoder_option_shapes(cin, hd, k, bn):
    s, ch, idx = {}, [cin, hd, hd * 2], 0
    for li in range(2):
        s[f"encoder.{idx}.weight"] = (ch[li + 1], ch[li], k, k)
        if bn:
            for nm in ("weight", "bias", "running_mean", "running_var"):
                s[f"encoder.{idx + 1}.{nm}"] = (ch[li + 1],)
            s[f"encoder.{idx + 1}.num_batches_tracked"] = ()
        else:
            s[f"encoder.{idx}.bias"] = (ch[li + 1],)
        idx += (3 if bn else 2) + 1
    return s


@pytest.mark.parametrize("tag,k,pad,bn,size", OPTION_CASES)
def test_encoder_constructor_options(tag, k, pad, bn, size):
    """CNNEncoder(kernel_size, padding, use_batchnorm) (src/model/cnn_encoder.py:16-18,44-61): the
    oracle against the reference's outputs, gradients and BatchNorm buffers."""
    g = golden("options")
    sd = gstate(encoder_option_shapes(3, 8, k, bn), "encopt_" + tag)
    x0 = G.gold_tensor(f"encopt/x_{tag}", (2, 3, size, size))
    y, _ = O.encoder_forward(x0, sd, prefix="", training=False, padding=pad, use_batchnorm=bn)
    close(y, g[f"enc_{tag}_y_eval"])
    for key in sd:
        if sd[key].dtype.is_floating_point and "running" not in key:
            sd[key].requires_grad_(True)
    x = x0.clone().requires_grad_(True)
    y, upd = O.encoder_forward(x, sd, prefix="", training=True, padding=pad, use_batchnorm=bn)
    close(y, g[f"enc_{tag}_y"])
    (y * G.gold_tensor(f"encopt/gy_{tag}", tuple(y.shape))).sum().backward()
    close_robust(x.grad, g[f"enc_{tag}_gx"], 2e-5)
    for key, v in upd.items():
        if "running" in key:
            close(v, g[f"enc_{tag}_buf_{key}"])
    for key in sd:
        if sd[key].requires_grad:
            close(sd[key].grad, g[f"enc_{tag}_g_{key}"], 1e-4)


def test_discriminator_three_input_channels():
    """PatchDiscriminator(input_channels=3) (src/model/gan.py:10,25)."""
    g = golden("options")
    sd = gstate(disc_shapes(16, 3), "disc3")
    for key in sd:
        if sd[key].dtype.is_floating_point and "running" not in key:
            sd[key].requires_grad_(True)
    x = G.gold_tensor("disc3/x", (2, 3, 50, 50)).requires_grad_(True)
    y, upd = O.discriminator_forward(x, sd)
    close(y, g["disc3_y"], 2e-5)
    (y * G.gold_tensor("disc3/gy", tuple(y.shape))).sum().backward()
    close_robust(x.grad, g["disc3_gx"], 1e-4)
    for key, v in upd.items():
        if "running" in key:
            close(v, g[f"disc3_buf_{key}"])
    for key in sd:
        if sd[key].requires_grad and key not in ("model.2.0.bias", "model.3.0.bias", "model.4.0.bias"):
            close(sd[key].grad, g[f"disc3_g_{key}"], 1e-3)


def test_calculate_metrics_resizes_targets():
    """src/loss.py:254-271: 4-D / 3-D targets of another shape are resized bilinearly first."""
    g = golden("metrics_resize")
    keys = ("RMSE", "MAE", "R2", "NSE", "MAPE", "KGE")
    p4 = G.gold_tensor("mres/p4", (3, 2, 40, 48)) + 2.0
    p3 = G.gold_tensor("mres/p3", (2, 30, 50)) + 1.5
    for tag, p, t in (("m4", p4, G.gold_tensor("mres/t4", (3, 2, 20, 24)) + 2.0),
                      ("m3", p3, G.gold_tensor("mres/t3", (2, 12, 20)) + 1.5),
                      ("m4_down", p4, G.gold_tensor("mres/t4d", (3, 2, 100, 96)) + 2.0)):
        m = O.calculate_metrics(p.numpy(), t.numpy())
        np.testing.assert_allclose([m[k] for k in keys], g[tag], rtol=1e-5, atol=1e-7)
