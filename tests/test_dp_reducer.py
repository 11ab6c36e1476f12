"""Data-parallel gradient exchange on CPU: two gloo ranks, the layout of training.FlatAdam
(small gradients gathered into one flat buffer, big ones reduced in place from a
post-accumulate hook) as training.Trainer uses it (src/train.py has no DP path; this is row (e)
of the hot-path contract)."""
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


def _net():
    torch.manual_seed(0)
    return nn.Sequential(nn.Linear(7, 130), nn.Linear(130, 5), nn.Linear(5, 3))


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sf = load_pkg()
        import importlib
        training = importlib.import_module(sf.__name__ + ".training")
        net = _net()
        opt = training.FlatAdam(net, lr=1e-3, big_numel=900)   # the 7x130 weight counts as big
        assert [p.numel() for p in opt.big] == [910]
        # parameters are views into the flat buffer, 64-float aligned, big ones last; no .grad yet
        for p, (o, n) in zip(opt.params, opt.ranges):
            assert o % 64 == 0 and p.data.data_ptr() == opt.flat[o:o + n].data_ptr() and p.grad is None
        assert opt.range_of(opt.big[0])[0] == opt.n_small
        pending = []

        def hook(param):
            pending.append(dist.all_reduce(param.grad, async_op=True))

        opt.big[0].register_post_accumulate_grad_hook(hook)
        x = torch.full((4, 7), float(rank + 1))
        net(x).sum().backward()          # the hook fires for the big weight during backward
        assert len(pending) == 1
        opt.gather()
        pending.append(dist.all_reduce(opt.gsmall, async_op=True))
        for w in pending:
            w.wait()
        out[rank] = (opt.gsmall.clone(), opt.big[0].grad.clone(), [r for r in opt.ranges])
        opt.zero_grad()
        assert all(p.grad is None for p in net.parameters())
    finally:
        dist.destroy_process_group()


def test_two_rank_gradient_reduce():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    (s0, b0, ranges), (s1, b1, _) = out[0], out[1]
    assert torch.equal(s0, s1) and torch.equal(b0, b1)   # every element reduced, on both ranks alike
    # expected: the sum over ranks of the single-process gradients
    net = _net()
    params = list(net.parameters())
    want = [torch.zeros_like(p) for p in params]
    for r in range(world):
        net.zero_grad()
        net(torch.full((4, 7), float(r + 1))).sum().backward()
        for w, p in zip(want, params):
            w += p.grad
    small = [w for w in want if w.numel() < 900]
    big = [w for w in want if w.numel() >= 900]
    torch.testing.assert_close(b0, big[0], rtol=1e-6, atol=1e-6)
    covered = 0
    for w, (o, n) in zip(small, ranges):
        assert n == w.numel()
        torch.testing.assert_close(s0[o:o + n].view(w.shape), w, rtol=1e-6, atol=1e-6)
        covered += n
    assert float(s0.abs().sum()) > 0 and covered == sum(w.numel() for w in small)


def test_gather_zero_fills_missing_gradients():
    """A parameter that received no gradient contributes zeros, not last step's values."""
    import importlib
    sf = load_pkg()
    training = importlib.import_module(sf.__name__ + ".training")
    net = _net()
    opt = training.FlatAdam(net, lr=1e-3)
    net(torch.ones(2, 7)).sum().backward()
    opt.gather()
    assert float(opt.gsmall.abs().sum()) > 0
    opt.zero_grad()
    net[2].weight.grad = torch.ones_like(net[2].weight)
    opt.gather()
    o, n = opt.range_of(net[2].weight)
    assert float(opt.gsmall[o:o + n].sum()) == n and float(opt.gsmall.abs().sum()) == n
