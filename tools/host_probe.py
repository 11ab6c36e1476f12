"""Host-side cost of a training step: time to enqueue the steps vs time until the GPU is done,
and a cProfile of the enqueue loop."""
import cProfile
import importlib
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

PKG = bench.PKG
sf = importlib.import_module(PKG)
training = importlib.import_module(PKG + ".training")
synth = importlib.import_module(PKG + ".synth")
ops = sf.ops
cfg = bench.bench_config(32)
torch.manual_seed(0)
region = cfg["data"]["region"]


def l2p(lon, lat):
    px, py = ops.lonlat_to_pixel(torch.from_numpy(lon).cuda(), torch.from_numpy(lat).cuda(), region, 200, 200)
    return px.cpu().numpy(), py.cpu().numpy()


batches = synth.make_batches(3, 32, seed=0, lonlat_to_pixel=l2p)
dev = [{"images": b["images"].cuda(), "output_image": b["output_image"].cuda(),
        "station_positions": b["station_positions"], "stations": b["stations"]} for b in batches]
tr = training.Trainer(cfg, device="cuda:0")


def step(i):
    b = dev[i % 3]
    return tr.train_step(b["images"], b["output_image"], b["station_positions"], b["stations"])


for i in range(3):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(12):
    step(3 + i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3 * (t1 - t0) / 12:.2f} ms/step, done {1e3 * (t2 - t0) / 12:.2f} ms/step")
pr = cProfile.Profile()
pr.enable()
for i in range(6):
    step(15 + i)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
