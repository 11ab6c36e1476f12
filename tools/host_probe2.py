"""Host time of a train step: how long the host takes to enqueue steps (no synchronisation inside)
against how long the GPU takes to run them."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
sf = importlib.import_module(bench.PKG)
training = importlib.import_module(bench.PKG + ".training")
synth = importlib.import_module(bench.PKG + ".synth")
cfg = bench.bench_config(32)
def l2p(lon, lat):
    px, py = sf.ops.lonlat_to_pixel(torch.from_numpy(lon).cuda(), torch.from_numpy(lat).cuda(), cfg["data"]["region"], 200, 200)
    return px.cpu().numpy(), py.cpu().numpy()
batches = synth.make_batches(3, 32, seed=0, lonlat_to_pixel=l2p)
dev = [{"images": b["images"].cuda(), "output_image": b["output_image"].cuda(), "station_positions": b["station_positions"], "stations": b["stations"]} for b in batches]
reh = os.environ.get("REH") == "1"
if reh:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
tr = training.Trainer(cfg, device="cuda", distributed=True if reh else None)
def step(i):
    b = dev[i % 3]
    tr.train_step(b["images"], b["output_image"], b["station_positions"], b["stations"])
TIMES = {}
def timed(obj, name, label=None):
    fn = getattr(obj, name)
    label = label or name
    def w(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            TIMES[label] = TIMES.get(label, 0.0) + time.perf_counter() - t
    (setattr if isinstance(obj, type) else object.__setattr__)(obj, name, w)
if os.environ.get("TIMERS") == "1":
    for n in ("_all_reduce", "_wait_reduces", "_step"):
        timed(tr, n)
    timed(tr.opt_g, "gather", "gather_g")
    timed(torch.Tensor, "backward")
    timed(tr.model, "forward", "model_forward")
    if tr.use_gan:
        timed(tr.discriminator, "forward", "disc_forward")
        timed(tr.opt_d, "gather", "gather_d")
        timed(tr.opt_d, "zero_grad", "zero_grad_d")
        timed(tr.gan_loss_fn, "gan_loss")
        timed(tr.gan_loss_fn, "discriminator_loss")
        timed(tr, "_set_d_requires_grad")
    timed(tr.loss_fn, "forward", "loss_forward")
    timed(sf.ops, "nan_inf_flag")
    timed(sf.ops, "image_station_loss")
    timed(importlib.import_module(bench.PKG + ".functional"), "pack_stations")
    timed(torch.Tensor, "to", "Tensor.to")
    timed(tr, "train_step")
for i in range(4): step(i)
TIMES.clear()
torch.cuda.synchronize()
N = 12
t0 = time.perf_counter()
for i in range(N): step(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
if TIMES:
    print({k: round(1e3 * v / N, 3) for k, v in TIMES.items()}, "ms/step on the host")
if os.environ.get("PROFILE_HOST") == "1":
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for i in range(N): step(i)
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats(os.environ.get("SORT", "cumulative")).print_stats(25)
print(f"host enqueue {1e3 * (t1 - t0) / N:.2f} ms/step, with GPU drain {1e3 * (t2 - t0) / N:.2f} ms/step")
