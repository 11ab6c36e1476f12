"""Data-parallel train step on the GPU: two ranks (one process each, both on cuda:0, gloo
carrying the gradients through host memory) run training.Trainer.train_step with the exchange path the
multi-GPU bench uses: big gradients all-reduced in place from post-accumulate hooks during
backward, all other gradients in one flat buffer, the 1/world scale folded into the Adam launch.
src/train.py has no data-parallel path; the properties checked are those of synchronous
gradient averaging."""
import importlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import goldgen as G
from conftest import close, load_pkg
from test_gpu_parity import load_gold, make_batches, train_cfg

pytestmark = pytest.mark.gpu
STEPS = 4


def _cfg(amp=False):
    cfg = train_cfg("Transformer", True, 1)
    cfg["system"]["big_grad_numel"] = 50000   # the temporal input Linear weight takes the hook path
    if amp:
        cfg["system"]["mixed_precision"] = True
    return cfg


def _run(sf, distributed, batches, shard=False, amp=False):
    training = importlib.import_module(sf.__name__ + ".training")
    cfg = _cfg(amp)
    cfg["system"]["dp_shard_big"] = shard
    tr = training.Trainer(cfg, device="cuda", distributed=distributed)
    assert tr.shard_big == (shard and distributed)
    assert len(tr.opt_g.big) >= 1 and len(tr.opt_g.small) >= 10
    # state right after construction: with ranks seeded differently it must already be rank 0's
    init = (tr.opt_g.flat.cpu().clone(), tr.opt_g.big[0].detach().cpu().clone(), tr.opt_d.flat.cpu().clone())
    load_gold(tr.model, "trn_g_trgan")
    load_gold(tr.discriminator, "trn_d_trgan")
    tr.model.temporal_module.dropout = 0.0
    losses = []
    for b in batches:
        o = tr.train_step(b["images"], b["output_image"], b["station_positions"], b["stations"])
        losses.append(float(o["total"].detach()))
    if shard and distributed:
        # before the gather a rank holds only its own slice of a big parameter up to date in f32 ...
        lo, hi = tr._shard_range(tr.opt_g.big[0])
        tr._wait_reduces()
        before = tr.opt_g.big[0].detach().reshape(-1).clone()
        shadow = tr.opt_g.big[0]._sf_shadow.float().reshape(-1).clone()
        tr.consolidate()
        after = tr.opt_g.big[0].detach().reshape(-1)
        assert torch.equal(before[lo:hi], after[lo:hi]) and (tr.world == 1 or not torch.equal(before, after))
        # ... while the bf16 shadow the forward pass reads is whole
        assert torch.equal(shadow, after.to(torch.bfloat16).float())
    torch.cuda.synchronize()
    assert not tr.saw_nan_or_inf()
    return tr.opt_g.flat.cpu(), tr.opt_d.flat.cpu(), losses, init


def _worker(rank, world, port, same_data, out, shard=False, amp=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    # gloo is a host transport: the trainer stages the gradients through host memory for it
    # (training.Trainer._all_reduce), exactly the calls, order and buffers the RCCL path uses
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sf = load_pkg()
        torch.manual_seed(1234 + rank)       # different random initialisation per rank
        batches = make_batches(2 * STEPS, 2, 3, 64)
        mine = batches[:STEPS] if (same_data or rank == 0) else batches[STEPS:]
        out[rank] = _run(sf, True, mine, shard, amp)
    finally:
        dist.destroy_process_group()


def _spawn(same_data, shard=False, amp=False):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = mp.Manager().dict()
    mp.spawn(_worker, args=(2, port, same_data, out, shard, amp), nprocs=2, join=True)
    return out[0], out[1]


def test_two_ranks_stay_in_lockstep_on_different_data():
    """Every gradient element is exchanged: after four steps (discriminator batches included)
    on different data both ranks hold bit-identical generator and discriminator parameters."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    (g0, d0, l0, i0), (g1, d1, l1, i1) = _spawn(same_data=False)
    assert all(torch.equal(a, b) for a, b in zip(i0, i1))   # rank 0's initial state was broadcast
    assert torch.equal(g0, g1) and torch.equal(d0, d1)
    assert l0 != l1                      # the ranks did see different batches
    assert np.isfinite(l0 + l1).all()


def test_two_ranks_on_the_same_data_match_one_process():
    """The average of two equal gradients is that gradient: the two-rank run must reproduce the
    single-process run (same tolerance as the reference-trajectory test: f32 atomics order)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    (g0, d0, l0, _), (g1, d1, _, _) = _spawn(same_data=True)
    assert torch.equal(g0, g1) and torch.equal(d0, d1)
    gs, ds, ls, _ = _run(load_pkg(), False, make_batches(2 * STEPS, 2, 3, 64)[:STEPS])
    np.testing.assert_allclose(l0, ls, rtol=5e-4)
    close(g0, gs.numpy(), 2e-3)
    close(d0, ds.numpy(), 2e-3)


def test_sharded_big_parameters_match_the_all_reduce_exchange():
    """system.dp_shard_big: reduce-scatter of the big gradient, Adam on the rank's slice, all-gather
    of the bf16 shadow.  After four steps on different data and the gather of the f32 slices both
    ranks hold bit-identical parameters, and they equal the all-reduce run's up to the f32 atomics
    order of the Linear layers (mixed precision, which the sliced update needs: 2e-3 on the
    losses, 5e-3 of the maximum on the parameters)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    (g0, d0, l0, _), (g1, d1, l1, _) = _spawn(same_data=False, shard=True, amp=True)
    assert torch.equal(g0, g1) and torch.equal(d0, d1)
    (ga, da, la, _), _ = _spawn(same_data=False, shard=False, amp=True)
    np.testing.assert_allclose(l0, la, rtol=2e-3)
    close(g0, ga.numpy(), 5e-3)
    close(d0, da.numpy(), 5e-3)


def _rccl_worker(rank, port, out, shard=False, amp=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        sf = load_pkg()
        torch.manual_seed(1234)
        out[0] = _run(sf, True, make_batches(2 * STEPS, 2, 3, 64)[:STEPS], shard, amp)
    finally:
        dist.destroy_process_group()


def test_rccl_exchange_path_with_one_rank_matches_the_plain_trainer():
    """The code path the multi-GPU runs take -- RCCL process group, hooks issuing the big
    all-reduces from the backward pass onto the communication stream, the flat-buffer reduces,
    the stream waits -- with a world of one rank, where the sum over ranks is the identity: the
    result must equal the trainer without any exchange."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = mp.Manager().dict()
    mp.spawn(_rccl_worker, args=(port, out), nprocs=1, join=True)
    g0, d0, l0, _ = out[0]
    torch.manual_seed(1234)
    gs, ds, ls, _ = _run(load_pkg(), False, make_batches(2 * STEPS, 2, 3, 64)[:STEPS])
    np.testing.assert_allclose(l0, ls, rtol=5e-4)
    close(g0, gs.numpy(), 2e-3)
    close(d0, ds.numpy(), 2e-3)


def test_rccl_sharded_exchange_with_one_rank_matches_the_plain_trainer():
    """The same with system.dp_shard_big: RCCL's reduce_scatter_tensor / all_gather_into_tensor on the
    communication stream and the MAX all-reduce of the step guard, with a world of one rank."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = mp.Manager().dict()
    mp.spawn(_rccl_worker, args=(port, out, True, True), nprocs=1, join=True)
    g0, d0, l0, _ = out[0]
    torch.manual_seed(1234)
    gs, ds, ls, _ = _run(load_pkg(), False, make_batches(2 * STEPS, 2, 3, 64)[:STEPS], amp=True)
    np.testing.assert_allclose(l0, ls, rtol=2e-3)
    close(g0, gs.numpy(), 5e-3)
    close(d0, ds.numpy(), 5e-3)
