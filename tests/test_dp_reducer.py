"""Data-parallel gradient exchange on CPU: two gloo ranks, the flat-buffer layout of
training.FlatAdam and the early-hook + remaining-ranges reduction of training.Trainer
(src/train.py has no DP path; this is row (e) of the hot-path contract)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from conftest import load_pkg


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sf = load_pkg()
        import importlib
        training = importlib.import_module(sf.__name__ + ".training")
        torch.manual_seed(0)
        net = nn.Sequential(nn.Linear(7, 130), nn.Linear(130, 5), nn.Linear(5, 3))
        opt = training.FlatAdam(net, lr=1e-3)
        # parameters and gradients are views into the flat buffers, 64-float aligned
        for p, (o, n) in zip(opt.params, opt.ranges):
            assert o % 64 == 0 and p.data.data_ptr() == opt.flat[o:o + n].data_ptr()
            assert p.grad.data_ptr() == opt.grad[o:o + n].data_ptr()
        big = net[0].weight
        bo, bn = opt.range_of(big)
        pending = []

        def hook(param):
            pending.append(dist.all_reduce(opt.grad[bo:bo + bn], async_op=True))

        big.register_post_accumulate_grad_hook(hook)
        x = torch.full((4, 7), float(rank + 1))
        net(x).sum().backward()          # the hook fires for the big weight during backward
        assert len(pending) == 1
        pending += training.reduce_ranges(opt.grad, [(bo, bn)])
        for w in pending:
            w.wait()
        out[rank] = opt.grad.clone()
    finally:
        dist.destroy_process_group()


def test_two_rank_flat_gradient_reduce():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    g0, g1 = out[0], out[1]
    assert torch.equal(g0, g1)           # every element reduced, none twice on one rank only
    # expected: the sum over ranks of the single-process gradients
    torch.manual_seed(0)
    net = nn.Sequential(nn.Linear(7, 130), nn.Linear(130, 5), nn.Linear(5, 3))
    want = [torch.zeros_like(p) for p in net.parameters()]
    for r in range(world):
        net.zero_grad()
        net(torch.full((4, 7), float(r + 1))).sum().backward()
        for w, p in zip(want, net.parameters()):
            w += p.grad
    pos = 0
    for w in want:
        n = w.numel()
        torch.testing.assert_close(g0[pos:pos + n].view(w.shape), w, rtol=1e-6, atol=1e-6)
        pos += (n + 63) // 64 * 64
    assert pos == g0.numel()


def test_reduce_ranges_cuts():
    """reduce_ranges covers exactly the complement of the skipped (alignment-rounded) ranges."""
    import importlib
    sf = load_pkg()
    training = importlib.import_module(sf.__name__ + ".training")
    calls = []

    class W:
        def wait(self):
            pass

    def fake(t, async_op=False):
        calls.append((t.storage_offset(), t.numel()))
        return W()

    real = dist.all_reduce
    dist.all_reduce = fake
    try:
        g = torch.zeros(1024)
        training.reduce_ranges(g, [(128, 100), (512, 64)])
        assert calls == [(0, 128), (256, 256), (576, 448)]
        calls.clear()
        training.reduce_ranges(g, [(0, 64)])
        assert calls == [(64, 960)]
        calls.clear()
        training.reduce_ranges(g, [])
        assert calls == [(0, 1024)]
    finally:
        dist.all_reduce = real
