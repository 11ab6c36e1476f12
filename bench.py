"""Benchmark of the training hot path: train samples/s at 10ch 200x200 T=5 (BASELINE.json).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one full iteration of the reference's batch loop (src/train.py:176-268) on one
resident synthetic batch per rank: generator forward, image+station loss, adversarial term,
generator backward through the discriminator, discriminator loss on every third batch, fused
Adam updates; with N > 1 ranks the gradients are averaged over RCCL.  Rank 0 prints ONE JSON
line.  The CPU oracle is timed beside it (cpu_baseline) on a bounded sample, rank 0, N == 1.
"""
import argparse
import importlib
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "tcn-transformer-gan-hybrid-model-for-predicting-streamflow-in-ungauged-basins_amd"

PMC_SUMMARY = "r2_pmc_hbm_traffic.json"
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak of MI355X (spec)


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


def cpu_baseline(cfg, batches, steps=3, batch=8):
    """The CPU oracle (oracle/streamflow_oracle.py, kind 'port') doing the identical step in f32
    on the host cores: `batch` samples per step, 1 warm-up + `steps` timed steps."""
    from oracle import streamflow_oracle as O
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import goldgen as G
    cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
    torch.set_num_threads(cores)
    sf = importlib.import_module(PKG)
    shapes_g = {k: tuple(v.shape) for k, v in sf.StreamflowPredictionModel(cfg).state_dict().items()}
    shapes_d = {k: tuple(v.shape) for k, v in sf.PatchDiscriminator(1, 64).state_dict().items()}
    keep = {k: O.positional_encoding(v[1], v[2]) for k, v in shapes_g.items() if k.endswith("pos_encoder.pe")}
    tr = O.TrainerOracle(cfg, G.gold_state(shapes_g, "bench_g", keep=keep), G.gold_state(shapes_d, "bench_d"))
    b0 = batches[0]
    args = (b0["images"][:batch], b0["output_image"][:batch], b0["station_positions"][:batch], b0["stations"][:batch])
    tr.step(*args)
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(*args)
    dt = time.perf_counter() - t0
    return {"value": round(batch * steps / dt, 4), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"{steps} timed steps (after 1 warm-up) of batch {batch}, same config and synthetic inputs, f32"}


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed counter passes over this same command
    (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs, summarised by
    tools/pmc_summary.py with the gfx950 doubling of FETCH_SIZE for wide reads).  Counters cannot
    be collected from inside the process, so the figure is the profiled one, not this run's."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", PMC_SUMMARY)
    try:
        with open(path) as f:
            row = json.load(f)["kernels"][kernel]
    except (OSError, KeyError, ValueError):
        return None, None
    return row["traffic_bytes_per_launch"], "profiles/" + PMC_SUMMARY


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
    ap.add_argument("--static-wgrad", action="store_true",
                    help="distributed / rehearsal runs: keep the weight-gradient kernel's fixed tile slices")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    rehearse = a.exchange_rehearsal and world == 1
    if rehearse:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local))
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    sf = importlib.import_module(PKG)
    from importlib import import_module
    training = import_module(PKG + ".training")
    synth = import_module(PKG + ".synth")
    ops = sf.ops

    cfg = bench_config(a.batch)
    if a.static_wgrad:
        cfg["system"]["dp_dynamic_wgrad"] = False
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

    def step(i):
        b = dev_batches[i % len(dev_batches)]
        return trainer.train_step(b["images"], b["output_image"], b["station_positions"], b["stations"])

    for i in range(a.warmup):
        ld = step(i)
    # HIP-event pairs cost GPU time, so inside the timed region only the launches of the dominant
    # kernel are bracketed (2 per step: the encoder's second conv and its data gradient); the other
    # MFMA kernels are timed in extra steps after it
    dom = "conv3_dma_wide_kernel"
    other_tags = ("conv3_dma_res_kernel", "conv3_dma_res_mask_kernel", "conv3_dma_kernel", "wgrad3_dma_kernel")
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ops.TIMED[dom] = []
    t0 = time.perf_counter()
    for i in range(a.steps):
        ld = step(a.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss = float(ld["total"].detach())
    bad = trainer.saw_nan_or_inf()
    dom_stats = ops.timed_tflops(dom)
    dom_bytes = ops.timed_bytes(dom)
    del ops.TIMED[dom]
    for t in other_tags:
        ops.TIMED[t] = []
    for i in range(3):
        step(a.warmup + a.steps + i)
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
        # shapes of the persistent forward / data-gradient kernel (res: weights resident in LDS), wgrad3_dma_kernel the weight gradient (its
        # event pair also spans the small slab-reduction launch that follows it).
        n_l, ms_l, achieved = dom_stats
        traffic, traffic_src = pmc_traffic(dom)
        others = {}
        for t in other_tags:
            n_o, ms_o, tf_o = other_stats[t]
            if n_o == 0:
                continue
            others[t] = {"launches": n_o, "ms_per_launch": round(ms_o, 4), "TFLOP/s": round(tf_o, 2),
                         "measured": "3 extra steps after the timed region"}
        out = {
            "metric": "train samples/sec (node) @10ch 200x200 T=5",
            "value": round(world * a.batch * a.steps / dt, 3), "unit": "samples/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "configs[2]/[3]: use_gan generator+PatchGAN adversarial step, 10ch 200x200 -> 400x400, "
                                   "T=5, Transformer, hidden 128, 64 stations k=5", "batch_per_gpu": a.batch,
                       "global_batch": a.batch * world, "parallelism": f"dp{world}"},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": BF16_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / BF16_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                         "traffic_unit": "bytes per launch", "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": round(dom_bytes), "kernel": dom, "launches": n_l, "ms_per_launch": round(ms_l, 4), "other": others},
            "final_loss": round(loss, 6), "nan_or_inf": bad,
        }
        if with_loader is not None:
            out["with_loader"] = with_loader
        if rehearse:
            out["exchange_rehearsal"] = True
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, batches)
        print(json.dumps(out))
    if world > 1 or rehearse:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
