"""Which torch (non-library) kernels run in a training step: torch.profiler table of one step."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
sf = importlib.import_module(bench.PKG)
training = importlib.import_module(bench.PKG + ".training")
synth = importlib.import_module(bench.PKG + ".synth")
ops = sf.ops
cfg = bench.bench_config(32)
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
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=False) as prof:
    for i in range(3, 6):
        step(i)
torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="count", row_limit=40, max_name_column_width=60))
