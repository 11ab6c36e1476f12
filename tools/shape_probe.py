"""Train steps at shapes the benchmark does not use: odd batch sizes, image sizes that are not
multiples of 8 / 4, few and many stations, LSTM and Transformer, f32 and bf16, GAN on / off.
Prints the loss after three steps; any kernel that assumes the benchmark shape fails here."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
sf = importlib.import_module(bench.PKG)
training = importlib.import_module(bench.PKG + ".training")

cases = [  # (H, W, batch, hidden, temporal, gan, amp, stations, up)
    (200, 200, 3, 128, "Transformer", True, True, 64, 2),
    (100, 140, 5, 64, "Transformer", True, True, 7, 2),
    (66, 50, 2, 32, "LSTM", True, False, 1, 2),
    (90, 90, 7, 64, "LSTM", False, True, 0, 1),
    (128, 128, 1, 128, "Transformer", True, True, 3, 2),
    (37, 53, 4, 16, "Transformer", True, True, 5, 3),
]
for (H, W, B, hid, temporal, gan, amp, S, up) in cases:
    cfg = {"data": {"image_size": [H, W], "window_size": 5, "upscale_factor": up, "region": [100, 110, 25, 35]},
           "model": {"input_channels": 10, "hidden_dim": hid, "num_layers": 2, "temporal_module": temporal,
                     "use_gan": gan, "gan_weight": 0.1},
           "train": {"lr": 1e-4}, "loss": {"metric": "MSE", "weights": {"image": 1.0, "station": 0.5}},
           "stations": {"kernel_size": 5, "metric": "MSE"}, "system": {"mixed_precision": amp, "grad_accum_steps": 1}}
    g = torch.Generator().manual_seed(0)
    tr = training.Trainer(cfg, device="cuda:0")
    img = torch.randn(B, 5, 10, H, W, generator=g).cuda()
    tgt = torch.randn(B, 1, H, W, generator=g).cuda()
    pos = [torch.stack([torch.randint(0, W, (S,), generator=g), torch.randint(0, H, (S,), generator=g)], dim=1) for _ in range(B)]
    run = [torch.randn(S, generator=g) for _ in range(B)]
    for _ in range(3):
        ld = tr.train_step(img, tgt, pos, run)
    torch.cuda.synchronize()
    ok = not tr.saw_nan_or_inf() and all(torch.isfinite(p).all() for p in tr.model.parameters())
    print(f"{H}x{W} B={B} hidden={hid} {temporal} gan={gan} amp={amp} S={S} up={up}: loss {float(ld['total'].detach()):.4f} finite={ok}", flush=True)
    assert ok
print("all shapes ok")
