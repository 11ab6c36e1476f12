"""Train-step and evaluate throughput of the other BASELINE configurations on one GPU:
configs[0] (128x128, LSTM, no GAN, f32, batch 4), configs[1] (128x128, Transformer, bf16, batch 32)
and configs[4] (batch inference at 200x200 -> 400x400, dates per second through Evaluator)."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
sf = importlib.import_module(bench.PKG)
training = importlib.import_module(bench.PKG + ".training")


def cfg(size, temporal, gan, amp, batch):
    return {"data": {"image_size": [size, size], "window_size": 5, "upscale_factor": 2, "region": [100, 110, 25, 35]},
            "model": {"input_channels": 10, "hidden_dim": 128, "num_layers": 2, "temporal_module": temporal,
                      "use_gan": gan, "gan_weight": 0.1},
            "train": {"lr": 1e-4, "batch_size": batch}, "loss": {"metric": "MSE", "weights": {"image": 1.0, "station": 0.5}},
            "stations": {"kernel_size": 5, "metric": "MSE"}, "system": {"mixed_precision": amp, "grad_accum_steps": 1}}


def train_rate(c, batch, size, steps=8):
    tr = training.Trainer(c, device="cuda:0")
    g = torch.Generator().manual_seed(0)
    img = torch.randn(batch, 5, 10, size, size, generator=g).cuda()
    tgt = torch.randn(batch, 1, size, size, generator=g).cuda()
    pos = [torch.randint(0, size, (5, 2), generator=g) for _ in range(batch)]
    run = [torch.randn(5, generator=g) for _ in range(batch)]
    for _ in range(3):
        ld = tr.train_step(img, tgt, pos, run)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ld = tr.train_step(img, tgt, pos, run)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return batch * steps / dt, float(ld["total"])


print("configs[0] LSTM f32 128x128 B=4 : %.1f samples/s (loss %.4f)" % train_rate(cfg(128, "LSTM", False, False, 4), 4, 128))
print("configs[1] Transformer bf16 128x128 B=32: %.1f samples/s (loss %.4f)" % train_rate(cfg(128, "Transformer", False, True, 32), 32, 128))
print("configs[2] f32 (no autocast) 200x200 GAN B=8: %.1f samples/s (loss %.4f)" % train_rate(cfg(200, "Transformer", True, False, 8), 8, 200))
# configs[4]: inference
c = cfg(200, "Transformer", False, False, 1)
model = sf.StreamflowPredictionModel(c).cuda()
ev = sf.Evaluator(c, model=model)
g = torch.Generator().manual_seed(1)
ds = [{"images": torch.randn(5, 10, 200, 200, generator=g), "output_image": torch.randn(1, 200, 200, generator=g),
       "stations": torch.randn(64, generator=g), "station_positions": torch.randint(0, 200, (64, 2), generator=g)}
      for _ in range(64)]
for amp in (False, True):
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
        ev.predict_batch(torch.stack([s["images"] for s in ds[:16]]))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for b0 in range(0, 64, 16):
            ev.predict_batch(torch.stack([s["images"] for s in ds[b0:b0 + 16]]))
        torch.cuda.synchronize()
        print("configs[4] forward only, batch 16, %s: %.1f dates/s" % ("bf16" if amp else "f32", 64 / (time.perf_counter() - t0)))
t0 = time.perf_counter()
res = ev.evaluate_dataset(ds, batch_size=16)
print("configs[4] evaluate_dataset incl. host metrics (SSIM etc.), f32: %.1f dates/s" % (64 / (time.perf_counter() - t0)))
