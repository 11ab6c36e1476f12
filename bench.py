"""Benchmark of the training hot path: train samples/s at 10ch 200x200 T=5 (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

One step = one full iteration of the reference's batch loop (src/train.py:176-268) on one
resident synthetic batch per rank: generator forward, image+station loss, adversarial term,
generator backward through the discriminator, discriminator loss on every third batch, fused
Adam updates; with N > 1 ranks the gradients are averaged over RCCL.  Rank 0 prints ONE JSON
line.  The CPU oracle is timed beside it (cpu_baseline) on a bounded sample, rank 0, N == 1.

Launching.  With N > 1 the ranks are separate processes, one per GPU.  Started under
torch.distributed.run (RANK / WORLD_SIZE / LOCAL_RANK in the environment) this file is one rank.
Started plainly (`python bench.py --gpus N`, no WORLD_SIZE) it starts the N ranks itself: a child
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` of
this same file, created before this process has made any GPU call, whose rank 0's JSON line is
relayed; the exit code is non-zero when any rank fails.  `--shared-gpu` puts every rank on cuda:0
and exchanges gradients through the staged (gloo) transport: a rehearsal of the N > 1 code path
on a one-GPU box, not a measurement.
"""
import argparse
import importlib
import json
import os
import socket
import statistics
import subprocess
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "tcn-transformer-gan-hybrid-model-for-predicting-streamflow-in-ungauged-basins_amd"

PMC_TRAFFIC = "r3_pmc_hbm_traffic.json"
PMC_MFMA = "r3_pmc_mfma.json"
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak of MI355X (spec)
REPEATS = 3                     # the timed block of K steps is run this many times; the median is reported


def bench_config(batch):
    return {
        "data": {"image_size": [200, 200], "window_size": 5, "upscale_factor": 2, "region": [100, 110, 25, 35]},
        "model": {"input_channels": 10, "hidden_dim": 128, "num_layers": 2, "temporal_module": "Transformer",
                  "use_gan": True, "gan_weight": 0.1},
        "train": {"lr": 1e-4, "batch_size": batch},
        "loss": {"metric": "MSE", "weights": {"image": 1.0, "station": 0.5}},
        "stations": {"kernel_size": 5, "metric": "MSE"},
        "system": {"mixed_precision": True, "grad_accum_steps": 1},
    }


def host_cores():
    """CPU threads this process may use: the scheduler affinity, capped by the harness's
    OMP_NUM_THREADS share when one is set (os.cpu_count() reports the whole machine)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    share = os.environ.get("OMP_NUM_THREADS")
    return max(1, min(n, int(share)) if share and share.isdigit() else n)


def cpu_baseline(cfg, batches, batch):
    """The CPU oracle (oracle/streamflow_oracle.py, kind 'port') doing the identical step in f32
    on the host cores: one warm-up step of a small batch (thread pool, allocator), then ONE timed
    step of the GPU run's own batch when the host has the memory for it (an f32 step of batch 32
    with autograd peaks at about 38 GiB), else three timed steps of batch 8."""
    from oracle import streamflow_oracle as O
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import goldgen as G
    cores = host_cores()
    torch.set_num_threads(cores)
    try:
        import psutil
        free_gb = psutil.virtual_memory().available / 2 ** 30
    except Exception:
        free_gb = 0.0
    full = free_gb >= 1.5 * batch + 16.0       # measured peak: 1.6 GiB + 1.14 GiB per sample
    nb, steps = (batch, 1) if full else (min(batch, 8), 3)
    sf = importlib.import_module(PKG)
    shapes_g = {k: tuple(v.shape) for k, v in sf.StreamflowPredictionModel(cfg).state_dict().items()}
    shapes_d = {k: tuple(v.shape) for k, v in sf.PatchDiscriminator(1, 64).state_dict().items()}
    keep = {k: O.positional_encoding(v[1], v[2]) for k, v in shapes_g.items() if k.endswith("pos_encoder.pe")}
    tr = O.TrainerOracle(cfg, G.gold_state(shapes_g, "bench_g", keep=keep), G.gold_state(shapes_d, "bench_d"))
    b0 = batches[0]

    def args(n):
        return (b0["images"][:n], b0["output_image"][:n], b0["station_positions"][:n], b0["stations"][:n])

    tr.step(*args(min(batch, 4)))
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(*args(nb))
    dt = time.perf_counter() - t0
    return {"value": round(nb * steps / dt, 4), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"{steps} timed step(s) of batch {nb} after a warm-up step of batch {min(batch, 4)}, same config "
                      f"and synthetic inputs, f32, {cores} threads ({free_gb:.0f} GiB of host memory free)"}


def profile_row(name, kernel):
    """Row of `kernel` in a committed counter summary under profiles/ (counters cannot be collected
    from inside the process: the figures are those of the profiled run of this same command)."""
    path = os.path.join(ROOT, "profiles", name)
    try:
        with open(path) as f:
            return json.load(f)["kernels"][kernel], "profiles/" + name
    except (OSError, KeyError, ValueError):
        return None, None


def self_launch(a):
    """Start the N ranks as children of a process that has not touched the GPU; relay rank 0's line."""
    if not a.shared_gpu and torch.cuda.device_count() < a.gpus:      # (counting devices initialises nothing)
        print(f"bench.py: --gpus {a.gpus} but {torch.cuda.device_count()} GPU(s) visible "
              f"(--shared-gpu rehearses the ranks on one GPU)", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for out in proc.stdout:
        try:
            if "metric" in json.loads(out):
                line = out.strip()
                continue
        except ValueError:
            pass
        sys.stderr.write(out)                   # anything else the ranks print is not the result
    rc = proc.wait()
    if rc != 0 or line is None:
        print(f"bench.py: the {a.gpus}-rank run failed (exit code {rc}, result line {'present' if line else 'missing'})",
              file=sys.stderr)
        return rc or 1
    print(line)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="samples per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exchange-rehearsal", action="store_true",
                    help="N=1 only: run the data-parallel gradient exchange (RCCL process group of one rank, hooks, "
                         "flat-buffer all-reduce) that N>1 uses, to rehearse that code on a one-GPU box")
    ap.add_argument("--shared-gpu", action="store_true",
                    help="N>1 only: every rank on cuda:0, gradients exchanged through host memory (gloo): a rehearsal "
                         "of the multi-rank path on a one-GPU box, not a measurement")
    ap.add_argument("--static-wgrad", action="store_true",
                    help="distributed / rehearsal runs: keep the weight-gradient kernel's fixed tile slices")
    ap.add_argument("--shard-big", action="store_true",
                    help="distributed runs: reduce-scatter the two big Linear gradients, Adam on each rank's slice, "
                         "all-gather the bf16 shadow (system.dp_shard_big) instead of the all-reduce")
    a = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        raise SystemExit(self_launch(a))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if a.shared_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    rehearse = a.exchange_rehearsal and world == 1
    if rehearse:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local))
    if world > 1:
        if a.shared_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    sf = importlib.import_module(PKG)
    training = importlib.import_module(PKG + ".training")
    synth = importlib.import_module(PKG + ".synth")
    ops = sf.ops

    cfg = bench_config(a.batch)
    if a.static_wgrad:
        cfg["system"]["dp_dynamic_wgrad"] = False
    if a.shard_big:
        cfg["system"]["dp_shard_big"] = True
    torch.manual_seed(0)
    region = cfg["data"]["region"]

    def l2p(lon, lat):
        px, py = ops.lonlat_to_pixel(torch.from_numpy(lon).cuda(), torch.from_numpy(lat).cuda(), region, 200, 200)
        return px.cpu().numpy(), py.cpu().numpy()

    # three resident batches so that the batch_idx % 3 discriminator schedule sees varied data
    batches = synth.make_batches(3, a.batch, seed=rank, lonlat_to_pixel=l2p)
    dev_batches = [{"images": b["images"].cuda(), "output_image": b["output_image"].cuda(),
                    "station_positions": b["station_positions"], "stations": b["stations"]} for b in batches]
    trainer = training.Trainer(cfg, device=f"cuda:{local}", distributed=True if rehearse else None)
    exchanging = trainer.distributed

    def step(i):
        b = dev_batches[i % len(dev_batches)]
        return trainer.train_step(b["images"], b["output_image"], b["station_positions"], b["stations"])

    for i in range(a.warmup):
        ld = step(i)
    # HIP-event pairs cost GPU time, so inside the timed region only the launches of the dominant
    # kernel are bracketed (2 per step: the encoder's second conv and its data gradient); the other
    # MFMA kernels are timed in extra steps after it
    dom = "conv3_dma_wide_kernel"
    other_tags = ("conv3_dma_res_kernel", "conv3_dma_res_mask_kernel", "conv3_dma_kernel", "wgrad_wide_kernel",
                  "wgrad3_dma_kernel")
    ops.TIMED[dom] = []
    if exchanging:
        trainer.comm_events = []
    # The timed block: EXACTLY K steps between barrier + synchronize on both sides, host clock and a
    # HIP-event pair on the compute stream.  It is run REPEATS times back to back (the first block
    # continues the schedule where the warm-up left it) and the median block is the one reported.
    wall, dev_ms = [], []
    nstep = a.warmup
    for _ in range(REPEATS):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        for i in range(a.steps):
            ld = step(nstep + i)
        e1.record()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        wall.append(time.perf_counter() - t0)
        dev_ms.append(e0.elapsed_time(e1))
        nstep += a.steps
    comm_ms = trainer.comm_exposed_ms() / (REPEATS * a.steps) if exchanging else None
    trainer.comm_events = None
    devices = [f"cuda:{torch.cuda.current_device()} {torch.cuda.get_device_name()}"]
    if world > 1:
        t = torch.tensor(wall + dev_ms, dtype=torch.float64, device="cpu" if a.shared_gpu else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)        # the slowest rank's clock, block by block
        wall, dev_ms = t[:REPEATS].tolist(), t[REPEATS:].tolist()
        gathered = [None] * world
        dist.all_gather_object(gathered, (devices[0], comm_ms))
        devices = [g[0] for g in gathered]
        comm_ms = max(g[1] for g in gathered)
    dt = statistics.median(wall)
    loss = float(ld["total"].detach())
    bad = trainer.saw_nan_or_inf()
    dom_stats = ops.timed_tflops(dom)
    dom_bytes = ops.timed_bytes(dom)
    del ops.TIMED[dom]
    for t in other_tags:
        ops.TIMED[t] = []
    for i in range(3):
        step(nstep + i)
    torch.cuda.synchronize()

    # second number: the same step fed by the device-resident dataset (window gather, station
    # packing and their host work inside the timed region), shuffled epochs over 3 batches' worth
    # of windows.  N=1 only.
    other_stats = {t: ops.timed_tflops(t) for t in other_tags}
    ops.TIMED.clear()
    with_loader = None
    if world == 1:
        raw, dates, st = synth.make_dataset_inputs(3 * a.batch + cfg["data"]["window_size"], seed=rank)
        ds = sf.ResidentStreamflowDataset.from_arrays(list(raw), dates, st, cfg)

        def epochs():
            e = 0
            while True:
                yield from ds.loader(a.batch, shuffle=True, seed=0, epoch=e, drop_last=True)
                e += 1

        feed = epochs()

        def loader_step():
            b = next(feed)
            return trainer.train_step(b["images"], b["output_image"], b["station_positions"], b["stations"])

        for _ in range(2):
            loader_step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(a.steps):
            loader_step()
        torch.cuda.synchronize()
        dt_l = time.perf_counter() - t1
        with_loader = {"value": round(a.batch * a.steps / dt_l, 3), "unit": "samples/s", "steps": a.steps,
                       "ms_per_step": round(dt_l / a.steps * 1e3, 3),
                       "loader": "ResidentStreamflowDataset.loader, shuffled, frames resident in HBM"}

    if rank == 0:
        # The three MFMA kernels of the 3x3 convolutions, every launch timed with HIP events on the
        # launch stream; algorithmic flops of a launch = 2 * N*H*W * Cin * Cout * 9 with the real
        # channel counts.  conv3_dma_wide_kernel / conv3_dma_kernel / conv3_dma_res_kernel / conv3_dma_res_mask_kernel are the
        # shapes of the persistent forward / data-gradient kernel (res: weights resident in LDS), wgrad_wide_kernel (more than 64 output channels) and wgrad3_dma_kernel the weight
        # gradient (their event pairs also span the small slab-reduction launch that follows).
        n_l, ms_l, achieved = dom_stats
        traffic_row, traffic_src = profile_row(PMC_TRAFFIC, dom)
        mfma_row, mfma_src = profile_row(PMC_MFMA, dom)
        others = {}
        for t in other_tags:
            n_o, ms_o, tf_o = other_stats[t]
            if n_o == 0:
                continue
            others[t] = {"launches": n_o, "ms_per_launch": round(ms_o, 4), "TFLOP/s": round(tf_o, 2),
                         "measured": "3 extra steps after the timed region"}
        ms_blocks = [w / a.steps * 1e3 for w in wall]
        out = {
            "metric": "train samples/sec (node) @10ch 200x200 T=5",
            "value": round(world * a.batch * a.steps / dt, 3), "unit": "samples/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "configs[2]/[3]: use_gan generator+PatchGAN adversarial step, 10ch 200x200 -> 400x400, "
                                   "T=5, Transformer, hidden 128, 64 stations k=5", "batch_per_gpu": a.batch,
                       "global_batch": a.batch * world, "parallelism": f"dp{world}"},
            "timing": {"blocks": REPEATS, "reported": "median block", "steps_per_block": a.steps,
                       "ms_per_step_host_clock": [round(m, 3) for m in ms_blocks],
                       "ms_per_step_hip_events": [round(m / a.steps, 3) for m in dev_ms],
                       "ms_per_step_min": round(min(ms_blocks), 3), "ms_per_step_max": round(max(ms_blocks), 3)},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": BF16_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / BF16_MFMA_PEAK_TFLOPS, 4),
                         "traffic": traffic_row["traffic_bytes_per_launch"] if traffic_row else None,
                         "traffic_unit": "bytes per launch", "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": round(dom_bytes), "kernel": dom, "launches": n_l,
                         "ms_per_launch": round(ms_l, 4),
                         "mfma_busy_frac": mfma_row.get("mfma_busy_frac") if mfma_row else None,
                         "eff_clock_ghz": mfma_row.get("eff_clock_ghz") if mfma_row else None,
                         "counters_source": mfma_src, "other": others},
            "devices": devices,
            "final_loss": round(loss, 6), "nan_or_inf": bad,
        }
        if exchanging:
            out["rccl_ranks"] = dist.get_world_size() if dist.get_backend() == "nccl" else 0
            out["comm_transport"] = "rccl" if dist.get_backend() == "nccl" else f"{dist.get_backend()} (staged through host memory)"
            out["comm_exposed_ms"] = round(comm_ms, 4)
            out["big_gradients"] = "reduce-scatter + sliced Adam + all-gather of the bf16 shadow" if trainer.shard_big else "all-reduce"
        if with_loader is not None:
            out["with_loader"] = with_loader
        if rehearse:
            out["exchange_rehearsal"] = True
        if a.shared_gpu and world > 1:
            out["shared_gpu_rehearsal"] = True
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, batches, a.batch)
        print(json.dumps(out), flush=True)
    if world > 1 or rehearse:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
