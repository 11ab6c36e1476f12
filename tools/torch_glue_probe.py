"""Lists the torch-native kernels of the bench's training step (the glue between the HIP calls) with
the Python frames that launch them: 6 profiled steps after 3 warm-up steps, counts per step."""
import importlib
import os
import sys
from collections import defaultdict
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
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
trainer = training.Trainer(cfg, device="cuda:0")


def step(i):
    b = dev[i % 3]
    return trainer.train_step(b["images"], b["output_image"], b["station_positions"], b["stations"])


for i in range(3):
    step(i)
torch.cuda.synchronize()
STEPS = 6
import traceback
from torch.utils._python_dispatch import TorchDispatchMode

LAUNCHING = ("copy_", "add_", "fill_", "zero_", "mul", "sum", "add", "div", "clone", "zeros", "_foreach", "_to_copy",
             "sub", "neg", "ones", "full", "cat", "index", "where", "mean", "sqrt", "max", "min", "clamp", "abs", "sigmoid",
             "zeros_like", "ones_like", "lerp", "addcmul", "addcdiv", "mul_", "div_", "sub_")
NOLAUNCH = ("view", "reshape", "detach", "empty", "as_strided", "expand", "permute", "transpose", "slice", "select",
            "unsqueeze", "squeeze", "alias", "_unsafe_view", "t.", "is_", "size", "stride", "numel", "item", "_local_scalar",
            "empty_like", "empty_strided", "new_empty", "split", "unbind", "narrow", "contiguous", "lift", "record_stream",
            "set_", "resize_", "is_pinned", "_pin_memory", "isfinite")
rows = defaultdict(int)


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).replace("aten.", "")
        if not any(name.startswith(k) for k in NOLAUNCH):
            frames = [f for f in traceback.extract_stack() if (PKG in f.filename or f.filename.endswith("bench.py"))]
            where = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in frames[-2:][::-1])
            big = max([a.numel() for a in args if isinstance(a, torch.Tensor)] + [0])
            rows[(name, where, "cuda" if any(isinstance(a, torch.Tensor) and a.is_cuda for a in args) or "cuda" in str(kwargs) else "cpu", big)] += 1
        return func(*args, **(kwargs or {}))


with Spy():
    for i in range(3, 3 + STEPS):
        step(i)
torch.cuda.synchronize()
for (name, where, devk, big), n in sorted(rows.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    print(f"{n / STEPS:6.2f}/step {devk:4s} {big:10d} {name:34s} {where}")
