"""Generate the golden vectors under tests/golden/ by running the REFERENCE on CPU.

Runs only where /root/reference exists (the build container).  It imports the reference
modules, feeds them formula-generated weights and inputs (tests/golden/goldgen.py) and
stores the reference's outputs as small .npz files.  No reference source is copied.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import os
import sys
import tempfile

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")

import numpy as np
import torch

import goldgen as G

from src.model import StreamflowPredictionModel  # noqa: E402
from src.model.cnn_encoder import CNNEncoder  # noqa: E402
from src.model.transformer_module import TransformerModule  # noqa: E402
from src.model.lstm_module import LSTMModule  # noqa: E402
from src.model.decoder import SimpleUpsamplingDecoder  # noqa: E402
from src.model.gan import PatchDiscriminator, GANLoss  # noqa: E402
from src.loss import CombinedLoss, StationLoss, ImageLoss, calculate_metrics  # noqa: E402
from src.dataset import StreamflowDataset  # noqa: E402

torch.set_num_threads(8)
torch.manual_seed(0)
np.random.seed(0)


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, {k: tuple(v.shape) for k, v in out.items()})


def load_gold(module, tag):
    sd = module.state_dict()
    gs = G.gold_state(G.shapes_of(sd), tag, keep=sd)
    module.load_state_dict(gs)
    return gs


def zero_dropout(model):
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0
        if isinstance(m, (torch.nn.LSTM, torch.nn.GRU)):
            m.dropout = 0.0


# ------------------------------------------------------------------ 1. pixel map
def gen_pixel():
    ds = StreamflowDataset.__new__(StreamflowDataset)
    out = {}
    for ci, (region, w, h) in enumerate(G.PIXEL_CASES):
        lon, lat = G.pixel_sweep(region, w, h)
        ds.region = list(region)
        ds.input_width, ds.input_height = w, h
        pts = [ds._lonlat_to_pixel(float(a), float(b)) for a, b in zip(lon, lat)]
        out[f"px{ci}"] = np.array([p[0] for p in pts], dtype=np.int32)
        out[f"py{ci}"] = np.array([p[1] for p in pts], dtype=np.int32)
    save("pixel_map", **out)


# ------------------------------------------------------------------ 2. encoder
def gen_encoder():
    out = {}
    for size in (64, 128, 200):
        enc = CNNEncoder(input_channels=4, hidden_dim=16, num_layers=2)
        load_gold(enc, "enc")
        x = G.gold_tensor(f"enc/x{size}", (2, 4, size, size))
        enc.train()
        out[f"train{size}"] = enc(x)
        sd = enc.state_dict()
        out[f"rm1_{size}"] = sd["encoder.1.running_mean"].clone()
        out[f"rv1_{size}"] = sd["encoder.1.running_var"].clone()
        out[f"rm2_{size}"] = sd["encoder.5.running_mean"].clone()
        out[f"rv2_{size}"] = sd["encoder.5.running_var"].clone()
        if size == 128:
            load_gold(enc, "enc")
            enc.eval()
            with torch.no_grad():
                out["eval128"] = enc(x)
    # backward through the train-mode encoder at 200 (overlapping pool bins)
    enc = CNNEncoder(input_channels=4, hidden_dim=16, num_layers=2)
    load_gold(enc, "enc")
    enc.train()
    x = G.gold_tensor("enc/x200", (2, 4, 200, 200)).requires_grad_(True)
    y = enc(x)
    gy = G.gold_tensor("enc/gy200", tuple(y.shape))
    (y * gy).sum().backward()
    out["g200_w0"] = enc.encoder[0].weight.grad
    out["g200_w4"] = enc.encoder[4].weight.grad
    out["g200_bn1w"] = enc.encoder[1].weight.grad
    out["g200_bn1b"] = enc.encoder[1].bias.grad
    out["g200_bn5w"] = enc.encoder[5].weight.grad
    out["g200_bn5b"] = enc.encoder[5].bias.grad
    out["g200_x_sub"] = x.grad[:, :, ::5, ::5]
    out["g200_x_sum"] = G.checksum({"x": x.grad})["x"]
    save("encoder", **out)


# ------------------------------------------------------------------ 3. temporal
def gen_temporal():
    out = {}
    B, T, D_in, Hd = 3, 5, 256, 32
    x0 = G.gold_tensor("tmp/x", (B, T, D_in))
    gy = G.gold_tensor("tmp/gy", (B, Hd))
    for kind in ("tr", "lstm"):
        if kind == "tr":
            mod = TransformerModule(D_in, Hd, num_layers=2, num_heads=4, dropout=0.0)
        else:
            mod = LSTMModule(D_in, Hd, num_layers=2, dropout=0.0)
        load_gold(mod, "tmp_" + kind)
        mod.train()
        x = x0.clone().requires_grad_(True)
        y = mod(x)
        (y * gy).sum().backward()
        out[f"{kind}_y"] = y
        out[f"{kind}_gx"] = x.grad
        for n, p in mod.named_parameters():
            out[f"{kind}_g_{n}"] = p.grad
        mod.eval()
        with torch.no_grad():
            out[f"{kind}_y_eval"] = mod(x0)
    save("temporal", **out)


# ------------------------------------------------------------------ 4. decoder
def gen_decoder():
    out = {}
    for (oh, ow) in ((64, 64), (100, 100)):
        dec = SimpleUpsamplingDecoder(16, oh, ow, hidden_channels=64)
        load_gold(dec, "dec")
        x = G.gold_tensor("dec/x", (2, 16, 32, 32)).requires_grad_(True)
        y = dec(x)
        gy = G.gold_tensor(f"dec/gy{oh}", tuple(y.shape))
        (y * gy).sum().backward()
        out[f"y{oh}"] = y
        out[f"gx{oh}"] = x.grad
        for n, p in dec.named_parameters():
            out[f"g{oh}_{n}"] = p.grad
    save("decoder", **out)


# ------------------------------------------------------------------ 5. discriminator
def gen_disc():
    out = {}
    for size in (64, 50):
        d = PatchDiscriminator(1, 16)
        load_gold(d, "disc")
        d.train()
        x = G.gold_tensor(f"disc/x{size}", (2, 1, size, size)).requires_grad_(True)
        y = d(x)
        gy = G.gold_tensor(f"disc/gy{size}", tuple(y.shape))
        (y * gy).sum().backward()
        out[f"y{size}"] = y
        out[f"gx{size}"] = x.grad
        sd = d.state_dict()
        for k in sd:
            if "running" in k:
                out[f"buf{size}_{k}"] = sd[k].clone()
        for n, p in d.named_parameters():
            out[f"g{size}_{n}"] = p.grad
    save("discriminator", **out)


# ------------------------------------------------------------------ 5b. constructor options
OPTION_CASES = (("nobn", dict(kernel_size=3, padding=1, use_batchnorm=False), 72),
                ("k4p2", dict(kernel_size=4, padding=2, use_batchnorm=True), 70),
                ("k1p0nobn", dict(kernel_size=1, padding=0, use_batchnorm=False), 64),
                ("k2p0", dict(kernel_size=2, padding=0, use_batchnorm=True), 90))


def gen_options():
    """Options the model class never passes but the constructors accept: CNNEncoder(kernel_size,
    padding, use_batchnorm) and PatchDiscriminator(input_channels != 1).  Train-mode forward, every
    parameter gradient, the input gradient and the BatchNorm buffers."""
    out = {}
    for tag, kw, size in OPTION_CASES:
        enc = CNNEncoder(input_channels=3, hidden_dim=8, num_layers=2, **kw)
        load_gold(enc, "encopt_" + tag)
        enc.train()
        x = G.gold_tensor(f"encopt/x_{tag}", (2, 3, size, size)).requires_grad_(True)
        y = enc(x)
        gy = G.gold_tensor(f"encopt/gy_{tag}", tuple(y.shape))
        (y * gy).sum().backward()
        out[f"enc_{tag}_y"] = y
        out[f"enc_{tag}_gx"] = x.grad
        for n, p in enc.named_parameters():
            out[f"enc_{tag}_g_{n}"] = p.grad
        for k, v in enc.state_dict().items():
            if "running" in k:
                out[f"enc_{tag}_buf_{k}"] = v.clone()
        load_gold(enc, "encopt_" + tag)
        enc.eval()
        with torch.no_grad():
            out[f"enc_{tag}_y_eval"] = enc(x.detach())
    d = PatchDiscriminator(3, 16)
    load_gold(d, "disc3")
    d.train()
    x = G.gold_tensor("disc3/x", (2, 3, 50, 50)).requires_grad_(True)
    y = d(x)
    gy = G.gold_tensor("disc3/gy", tuple(y.shape))
    (y * gy).sum().backward()
    out["disc3_y"] = y
    out["disc3_gx"] = x.grad
    for k, v in d.state_dict().items():
        if "running" in k:
            out[f"disc3_buf_{k}"] = v.clone()
    for n, p in d.named_parameters():
        out[f"disc3_g_{n}"] = p.grad
    save("options", **out)


# ------------------------------------------------------------------ 6. losses
def station_cases(H, W):
    pos = [
        torch.tensor([[0, 0], [W - 1, H - 1], [5, 7], [6, 7], [5, 7], [W - 1, 0], [1, H - 2]], dtype=torch.int64),
        torch.zeros((0, 2), dtype=torch.int64),
        torch.tensor([[10, 3], [11, 4], [W // 2, H // 2]], dtype=torch.int64),
    ]
    run = [G.gold_tensor("loss/r0", (7,)), torch.zeros((0,)), G.gold_tensor("loss/r2", (3,))]
    return pos, run


def gen_losses():
    out = {}
    B, H, W = 3, 40, 48
    pred0 = G.gold_tensor("loss/pred", (B, 1, H, W))
    tgt = G.gold_tensor("loss/tgt", (B, 1, H // 2, W // 2))
    tgt[:, :, :5, :7] = 0.0  # exact zeros -> weight 1 region
    tgt_same = G.gold_tensor("loss/tgt_same", (B, 1, H, W))
    tgt_same[:, :, 10:20, :] = 0.0
    pos, run = station_cases(H, W)
    for metric in ("MSE", "MAE"):
        for nm, t in (("up", tgt), ("same", tgt_same)):
            p = pred0.clone().requires_grad_(True)
            l = ImageLoss(metric=metric)(p, t)
            l.backward()
            out[f"img_{metric}_{nm}"] = l
            out[f"img_{metric}_{nm}_g"] = p.grad
        for k in (3, 5):
            p = pred0.clone().requires_grad_(True)
            l = StationLoss(kernel_size=k, metric=metric)(p, pos, run)
            l.backward()
            out[f"stn_{metric}_{k}"] = l
            out[f"stn_{metric}_{k}_g"] = p.grad
    cfg = {"loss": {"metric": "MSE", "weights": {"image": 1.0, "station": 0.5}},
           "stations": {"kernel_size": 5, "metric": "MSE"}, "data": {"upscale_factor": 2}}
    p = pred0.clone().requires_grad_(True)
    ld = CombinedLoss(cfg)(p, tgt, pos, run)
    ld["total"].backward()
    out["comb_total"], out["comb_image"], out["comb_station"] = ld["total"], ld["image"], ld["station"]
    out["comb_g"] = p.grad
    # no stations at all -> station term is a constant 0
    ld = CombinedLoss(cfg)(pred0, tgt, [torch.zeros((0, 2), dtype=torch.int64)] * B, [torch.zeros(0)] * B)
    out["comb_nostn_total"] = ld["total"]
    out["comb_nostn_station"] = ld["station"]
    gl = GANLoss(0.1, 1.0)
    lr = G.gold_tensor("loss/logit_real", (B, 1, 5, 6), 2.0)
    lf = G.gold_tensor("loss/logit_fake", (B, 1, 11, 12), 2.0)
    a = lf.clone().requires_grad_(True)
    l = gl.gan_loss(a, torch.ones_like(a))
    l.backward()
    out["gan_g"], out["gan_g_grad"] = l, a.grad
    a = lr.clone().requires_grad_(True)
    b = lf.clone().requires_grad_(True)
    l = gl.discriminator_loss(a, b)
    l.backward()
    out["gan_d"], out["gan_d_greal"], out["gan_d_gfake"] = l, a.grad, b.grad
    # metrics
    pv = G.gold_array("loss/mp", (200,), 1.0, np.float64) + 3.0
    tv = pv + G.gold_array("loss/mt", (200,), 0.5, np.float64)
    m = calculate_metrics(torch.from_numpy(pv).unsqueeze(-1), torch.from_numpy(tv).unsqueeze(-1))
    out["metrics"] = np.array([m[k] for k in ("RMSE", "MAE", "R2", "NSE", "MAPE", "KGE")])
    save("losses", **out)


def gen_metrics_resize():
    """calculate_metrics with a target of another shape (src/loss.py:254-271): 4-D and 3-D targets
    are resized bilinearly to the prediction's last two dimensions first."""
    keys = ("RMSE", "MAE", "R2", "NSE", "MAPE", "KGE")
    out = {}
    p4 = G.gold_tensor("mres/p4", (3, 2, 40, 48)) + 2.0
    t4 = G.gold_tensor("mres/t4", (3, 2, 20, 24)) + 2.0
    m = calculate_metrics(p4, t4)
    out["m4"] = np.array([m[k] for k in keys])
    p3 = G.gold_tensor("mres/p3", (2, 30, 50)) + 1.5
    t3 = G.gold_tensor("mres/t3", (2, 12, 20)) + 1.5
    m = calculate_metrics(p3, t3)
    out["m3"] = np.array([m[k] for k in keys])
    t4d = G.gold_tensor("mres/t4d", (3, 2, 100, 96)) + 2.0          # a larger target: resized DOWN
    m = calculate_metrics(p4, t4d)
    out["m4_down"] = np.array([m[k] for k in keys])
    save("metrics_resize", **out)


# ------------------------------------------------------------------ 7. full generator forward
def model_cfg(hidden, temporal, size, channels, use_gan=False, accum=1, k=3):
    return {
        "data": {"image_size": [size, size], "window_size": 5, "upscale_factor": 2,
                 "output_dir": tempfile.mkdtemp(prefix="gold_"), "region": [100, 110, 25, 35]},
        "model": {"input_channels": channels, "hidden_dim": hidden, "num_layers": 2,
                  "temporal_module": temporal, "use_gan": use_gan, "gan_weight": 0.1},
        "train": {"device": "cpu", "lr": 1e-3, "batch_size": 2},
        "loss": {"metric": "MSE", "weights": {"image": 1.0, "station": 0.5}},
        "stations": {"kernel_size": k, "metric": "MSE"},
        "system": {"mixed_precision": False, "grad_accum_steps": accum, "log_interval": 1000,
                   "use_rich": False},
    }


def gen_generator():
    out = {}
    for temporal in ("LSTM", "Transformer"):
        cfg = model_cfg(16, temporal, 64, 3)
        m = StreamflowPredictionModel(cfg)
        load_gold(m, "gen_" + temporal)
        zero_dropout(m)
        x = G.gold_tensor("gen/x", (2, 5, 3, 64, 64))
        m.eval()
        with torch.no_grad():
            out[f"{temporal}_eval"] = m(x)
        m.train()
        out[f"{temporal}_train"] = m(x)
    # full-size C3 shape: hidden 128, 10ch 200x200 -> 400x400, Transformer, one sample
    cfg = model_cfg(128, "Transformer", 200, 10)
    m = StreamflowPredictionModel(cfg)
    load_gold(m, "gen_full")
    m.eval()
    x = G.gold_tensor("gen_full/x", (1, 5, 10, 200, 200))
    with torch.no_grad():
        out["full_eval"] = m(x)
    save("generator", **out)


# ------------------------------------------------------------------ 8. train_epoch semantics
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


def gen_train():
    from src.train import Trainer
    out = {}
    for temporal, use_gan, accum, tag in (("Transformer", True, 2, "trgan"), ("LSTM", False, 1, "lstm")):
        cfg = model_cfg(16, temporal, 64, 3, use_gan=use_gan, accum=accum, k=5)
        tr = Trainer(cfg)
        load_gold(tr.model, "trn_g_" + tag)
        zero_dropout(tr.model)
        if use_gan:
            load_gold(tr.discriminator, "trn_d_" + tag)
        batches = make_batches(4, 2, 3, 64)
        losses = []
        g_init = {k: v.clone() for k, v in tr.model.state_dict().items()}
        d_init = {k: v.clone() for k, v in tr.discriminator.state_dict().items()} if use_gan else None

        class Rec(list):
            pass

        # wrap loss logging: run batches one at a time so per-batch loss dicts are visible
        orig = tr.loss_fn.forward
        rec = []

        def hook(*a, **kw):
            d = orig(*a, **kw)
            rec.append(d)
            return d

        tr.loss_fn.forward = hook
        res = tr.train_epoch(batches)
        for i, d in enumerate(rec):
            row = [float(d["total"]), float(d["image"]), float(d["station"]),
                   float(d.get("gan_g", torch.tensor(float("nan")))),
                   float(d.get("gan_d", torch.tensor(float("nan"))))]
            losses.append(row)
        out[f"{tag}_losses"] = np.array(losses, dtype=np.float64)
        out[f"{tag}_epoch"] = np.array([res["loss"], res["image_loss"], res["station_loss"]])
        gsd = tr.model.state_dict()
        cs = G.checksum(gsd, [k for k in gsd if "num_batches" not in k and "pos_encoder" not in k])
        out[f"{tag}_gkeys"] = np.array(list(cs.keys()))
        out[f"{tag}_gsum"] = np.stack(list(cs.values()))
        pj = G.update_projection(gsd, g_init, list(cs.keys()))
        out[f"{tag}_gproj"] = np.stack([pj[k] for k in cs])
        out[f"{tag}_g_dec_out_w"] = gsd["decoder.output_conv.weight"]
        out[f"{tag}_g_bn1_rm"] = gsd["cnn_encoder.encoder.1.running_mean"]
        if use_gan:
            dsd = tr.discriminator.state_dict()
            cs = G.checksum(dsd, [k for k in dsd if "num_batches" not in k])
            out[f"{tag}_dkeys"] = np.array(list(cs.keys()))
            out[f"{tag}_dsum"] = np.stack(list(cs.values()))
            pj = G.update_projection(dsd, d_init, list(cs.keys()))
            out[f"{tag}_dproj"] = np.stack([pj[k] for k in cs])
            out[f"{tag}_dsign"] = G.update_signs(dsd, d_init, list(cs.keys()))      # D takes one step in 4 batches
            out[f"{tag}_d_m5_b"] = dsd["model.5.bias"]
            out[f"{tag}_d_bn2_rv"] = dsd["model.2.1.running_var"]
    save("train", **out)


# ------------------------------------------------------------------ 9. evaluate path
def gen_evaluate():
    from generate_dummy_data import generate_dummy_data
    from src.evaluate import Evaluator
    import contextlib
    import io
    tmp = tempfile.mkdtemp(prefix="gold_eval_")
    np.random.seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        generate_dummy_data(days=9, stations=3, channels=2, height=64, width=64, output_dir=tmp)
    cfg = model_cfg(16, "LSTM", 64, 2, k=3)
    cfg["data"].update({"image_dir": os.path.join(tmp, "images"), "station_dir": os.path.join(tmp, "stations"),
                        "output_dir": os.path.join(tmp, "output"), "region": [90, 150, -15, 25]})
    with contextlib.redirect_stdout(io.StringIO()):
        ds = StreamflowDataset(cfg["data"]["image_dir"], cfg["data"]["station_dir"], cfg, normalize=True)
    m = StreamflowPredictionModel(cfg)
    load_gold(m, "eval_g")
    path = os.path.join(tmp, "model.pt")
    torch.save(m.state_dict(), path)
    ev = Evaluator(cfg, path)
    res = ev.evaluate_dataset(ds, os.path.join(tmp, "evalout"))
    out = {"n": np.array(len(ds))}
    for i in range(len(ds)):
        s = ds[i]
        out[f"images{i}"] = s["images"]
        out[f"target{i}"] = s["output_image"]
        out[f"stations{i}"] = s["stations"]
        out[f"pos{i}"] = s["station_positions"]
    out["metrics"] = np.array([res[k] for k in ("RMSE", "MAE", "SSIM", "PSNR", "Correlation")])
    sm = res["station_metrics"]
    out["station_metrics"] = np.array([sm[k] for k in ("RMSE", "MAE", "R2", "NSE", "MAPE", "KGE")])
    save("evaluate", **out)


def gen_evaluate_csv():
    """The two CSV files Evaluator.evaluate_dataset writes (src/evaluate.py:294-333) for the run of
    gen_evaluate: header and rows as text, plus the sample dates the rows carry."""
    from generate_dummy_data import generate_dummy_data
    from src.evaluate import Evaluator
    import contextlib
    import csv
    import io
    tmp = tempfile.mkdtemp(prefix="gold_evalcsv_")
    np.random.seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        generate_dummy_data(days=9, stations=3, channels=2, height=64, width=64, output_dir=tmp)
    cfg = model_cfg(16, "LSTM", 64, 2, k=3)
    cfg["data"].update({"image_dir": os.path.join(tmp, "images"), "station_dir": os.path.join(tmp, "stations"),
                        "output_dir": os.path.join(tmp, "output"), "region": [90, 150, -15, 25]})
    with contextlib.redirect_stdout(io.StringIO()):
        ds = StreamflowDataset(cfg["data"]["image_dir"], cfg["data"]["station_dir"], cfg, normalize=True)
    m = StreamflowPredictionModel(cfg)
    load_gold(m, "eval_g")
    path = os.path.join(tmp, "model.pt")
    torch.save(m.state_dict(), path)
    outdir = os.path.join(tmp, "evalout")
    Evaluator(cfg, path).evaluate_dataset(ds, outdir)
    out = {"dates": np.array([str(ds[i]["date"]) for i in range(len(ds))])}
    for name in ("stations_eval", "evaluation_metrics"):
        with open(os.path.join(outdir, name + ".csv"), newline="") as f:
            rows = list(csv.reader(f))
        out[name + "_header"] = np.array(rows[0])
        out[name + "_rows"] = np.array(rows[1:])
    out["files"] = np.array(sorted(os.listdir(outdir)))
    save("evaluate_csv", **out)


def gen_disc_eval():
    """PatchDiscriminator in eval mode (BatchNorm on its running statistics, src/model/gan.py:24-52)."""
    out = {}
    for size in (64, 50):
        d = PatchDiscriminator(1, 16)
        load_gold(d, "disc")
        d.eval()
        x = G.gold_tensor(f"disc/x{size}", (2, 1, size, size)).requires_grad_(True)
        y = d(x)
        gy = G.gold_tensor(f"disc/gy{size}", tuple(y.shape))
        (y * gy).sum().backward()
        out[f"y{size}"] = y
        out[f"gx{size}"] = x.grad
        out[f"g{size}_bn_w"] = d.model[3][1].weight.grad
        out[f"g{size}_conv_w"] = d.model[3][0].weight.grad
    save("discriminator_eval", **out)


def dataset_case(tag):
    """Raw inputs of one dataset case (formula/seeded; written to disk for the reference to read)."""
    rs = np.random.RandomState({"A": 11, "B": 12}[tag])
    if tag == "A":
        D, C, h, w, size, T, norm = 9, 3, 20, 24, [20, 24], 5, "z_score"
    else:
        D, C, h, w, size, T, norm = 8, 2, 12, 16, [20, 24], 3, "minmax"
    frames = rs.gamma(0.6, 2.0, size=(D, C, h, w)).astype(np.float32)   # many values below the 0.1 flow threshold
    frames[np.abs(np.abs(frames) - 0.1) < 1e-3] = 0.25                  # keep clear of the threshold itself
    frames[:, 1:] += rs.standard_normal((D, C - 1, h, w)).astype(np.float32) * 3.0
    if tag == "A":
        frames[1, 1, 3, 4] = np.nan          # an input-only channel
        frames[6, 0, 7, 7] = np.nan          # channel 0 of a date that is also a target
        frames[3, 2, 0, 0] = np.nan
    masks = {}
    for d in ([2, 6] if tag == "A" else [5]):
        masks[d] = (rs.uniform(size=(h, w)) > 0.3).astype(np.uint8)
    dates = [f"202001{d + 1:02d}" for d in range(D)]
    names = ["st_b", "st_a", "gauge10", "x9"]
    stations = {}
    for i, n in enumerate(names):
        ts = list(dates)
        lon = np.full(D, 100.3 + 2.9 * i) + (rs.uniform(size=D) * 3.0 if i == 2 else 0.0)   # one station moves
        lat = np.full(D, 34.6 - 2.6 * i)
        run = rs.gamma(2.0, 30.0, size=D)
        if i == 3:                                   # rows missing on some dates
            keep = [k for k in range(D) if k % 3 != 0]
            ts, lon, lat, run = [ts[k] for k in keep], lon[keep], lat[keep], run[keep]
        if i == 1:                                   # duplicate date: the first row wins
            ts = ts + [ts[-1]]
            lon, lat, run = np.append(lon, 101.0), np.append(lat, 30.0), np.append(run, 7.0)
        stations[n] = (np.array([int(t) for t in ts]), np.asarray(lon), np.asarray(lat), np.asarray(run))
    cfg = {"data": {"window_size": T, "upscale_factor": 2, "region": [100, 110, 25, 35], "flow_threshold": 0.1,
                    "image_size": size, "normalization_type": norm}}
    return frames, masks, dates, stations, cfg


def write_dataset_case(root, frames, masks, dates, stations):
    import pandas as pd
    os.makedirs(os.path.join(root, "images"))
    os.makedirs(os.path.join(root, "stations"))
    for d, name in enumerate(dates):
        np.save(os.path.join(root, "images", name + ".npy"), frames[d])
        if d in masks:
            np.save(os.path.join(root, "images", name + "_mask.npy"), masks[d])
    for n, (ts, lon, lat, run) in stations.items():
        pd.DataFrame({"timestamp": ts, "lon": lon, "lat": lat, "runoff": run}).to_csv(
            os.path.join(root, "stations", n + ".csv"), index=False)


def gen_dataset():
    """StreamflowDataset (src/dataset.py) on two small synthetic directories: statistics and every
    sample.  The raw inputs are stored too: the test rewrites the directory from them."""
    import contextlib
    import io
    out = {}
    for tag in ("A", "B"):
        frames, masks, dates, stations, cfg = dataset_case(tag)
        tmp = tempfile.mkdtemp(prefix="gold_ds_")
        write_dataset_case(tmp, frames, masks, dates, stations)
        cfg["data"]["output_dir"] = os.path.join(tmp, "output")
        with contextlib.redirect_stdout(io.StringIO()):
            ds = StreamflowDataset(os.path.join(tmp, "images"), os.path.join(tmp, "stations"), cfg, normalize=True)
        out[f"{tag}_frames"] = frames
        out[f"{tag}_mask_idx"] = np.array(sorted(masks), dtype=np.int64)
        out[f"{tag}_masks"] = np.stack([masks[d] for d in sorted(masks)])
        for n, (ts, lon, lat, run) in stations.items():
            out[f"{tag}_st_{n}"] = np.stack([ts.astype(np.float64), lon, lat, run])
        out[f"{tag}_order"] = np.array(list(ds.stations.keys()))
        for k in ("image_mean", "image_std", "image_min", "image_max"):
            out[f"{tag}_{k}"] = getattr(ds, k)
        out[f"{tag}_runoff_stats"] = np.array([ds.runoff_mean, ds.runoff_std, ds.runoff_min, ds.runoff_max])
        out[f"{tag}_n"] = np.array(len(ds))
        for i in range(len(ds)):
            smp = ds[i]
            date = smp["date"]
            out[f"{tag}_images{i}"] = smp["images"]
            out[f"{tag}_target{i}"] = smp["output_image"]
            out[f"{tag}_stations{i}"] = smp["stations"]
            out[f"{tag}_pos{i}"] = smp["station_positions"]
            out[f"{tag}_date{i}"] = np.array(date)
            out[f"{tag}_names{i}"] = np.array([n for n, df in ds.stations.items()
                                               if (df["timestamp"].astype(str) == date).any()])
    save("dataset", **out)


def gen_ckpt():
    """Key names, shapes and dtypes of the checkpoint the reference writes
    (torch.save(model.state_dict()), src/train.py:472-490) for the train-test configurations."""
    out = {}
    for tag, temporal in (("tr", "Transformer"), ("lstm", "LSTM")):
        m = StreamflowPredictionModel(model_cfg(16, temporal, 64, 3))
        sd = m.state_dict()
        out[f"{tag}_keys"] = np.array(list(sd.keys()))
        out[f"{tag}_shapes"] = np.array([list(v.shape) + [-1] * (5 - v.dim()) for v in sd.values()])
        out[f"{tag}_dtypes"] = np.array([str(v.dtype) for v in sd.values()])
    save("checkpoint_keys", **out)


def gen_baselines():
    """PureLSTMModel / PureBPModel (src/model/baseline_models.py): eval forward, train forward,
    gradients of sum(y * u) for every parameter (checksums; a few in full), BatchNorm buffers
    after the training forward.  Dropout is zeroed."""
    from src.model.baseline_models import build_baseline_model
    out = {}
    cfg = {"model": {"input_channels": 3, "hidden_dim": 32, "num_layers": 2},
           "data": {"image_size": [64, 64], "upscale_factor": 2, "window_size": 5}}
    x = G.gold_tensor("base/x", (2, 5, 3, 64, 64))
    u = G.gold_tensor("base/u", (2, 1, 128, 128))
    for name in ("LSTM", "BP", "Attention-GRU"):
        m = build_baseline_model(name, cfg)
        load_gold(m, "base_" + name)
        zero_dropout(m)
        m.eval()
        with torch.no_grad():
            out[f"{name}_eval"] = m(x)
        m.train()
        y = m(x)
        (y * u).sum().backward()
        out[f"{name}_train"] = y
        grads = {k: p.grad for k, p in m.named_parameters()}
        cs = G.checksum(grads)
        out[f"{name}_gkeys"] = np.array(list(cs.keys()))
        out[f"{name}_gsum"] = np.stack([cs[k] for k in cs])
        for k in ("reducer.encoder.0.weight", "reducer.encoder.6.weight", "reducer.encoder.4.weight", "decoder.0.bias",
                  "decoder.2.weight", "head_proj.bias"):
            out[f"{name}_g_{k}"] = grads[k]
        sd = m.state_dict()
        out[f"{name}_keys"] = np.array(list(sd.keys()))
        out[f"{name}_rm7"] = sd["reducer.encoder.7.running_mean"]
        out[f"{name}_rv1"] = sd["reducer.encoder.1.running_var"]
    save("baselines", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["pixel", "encoder", "temporal", "decoder", "disc", "losses", "generator",
                             "train", "evaluate", "dataset", "ckpt", "baselines", "evaluate_csv", "disc_eval", "options",
                             "metrics_resize"]
    fns = {"pixel": gen_pixel, "encoder": gen_encoder, "temporal": gen_temporal, "decoder": gen_decoder,
           "disc": gen_disc, "losses": gen_losses, "generator": gen_generator, "train": gen_train,
           "evaluate": gen_evaluate, "dataset": gen_dataset, "ckpt": gen_ckpt, "baselines": gen_baselines,
           "evaluate_csv": gen_evaluate_csv, "disc_eval": gen_disc_eval, "options": gen_options,
           "metrics_resize": gen_metrics_resize}
    for w in which:
        fns[w]()
