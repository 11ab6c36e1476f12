"""Host-side logic of the resident dataset that needs no GPU: epoch sharding and the packed
station buffer layout."""
import numpy as np
import torch

from conftest import load_pkg


def test_window_shards_partition_an_epoch():
    sf = load_pkg()
    n, bs, world = 103, 8, 4
    per_rank = [sf.window_shards(n, bs, True, 3, 5, r, world, False) for r in range(world)]
    flat = [np.concatenate(p) for p in per_rank]
    assert all(len(f) == n // world for f in flat)                      # equal step counts on every rank
    assert len(set(np.concatenate(flat))) == (n // world) * world       # disjoint
    assert all(len(b) == bs for b in per_rank[0][:-1]) and len(per_rank[0][-1]) == (n // world) % bs
    assert len(sf.window_shards(n, bs, True, 3, 5, 0, world, True)) == (n // world) // bs
    again = sf.window_shards(n, bs, True, 3, 5, 2, world, False)
    assert all(np.array_equal(a, b) for a, b in zip(again, per_rank[2]))       # same seed + epoch, same order
    other = sf.window_shards(n, bs, True, 3, 6, 2, world, False)
    assert not np.array_equal(np.concatenate(other), flat[2])                  # next epoch reshuffles
    seq = sf.window_shards(10, 4, False, 0, 0, 0, 1, False)
    assert [b.tolist() for b in seq] == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9]]


def test_packed_stations_layout():
    sf = load_pkg()
    import importlib
    SF = importlib.import_module(sf.__name__ + ".functional")
    pos = np.array([[3, 4], [5, 6], [7, 8]])
    run = np.array([0.5, -1.25, 2.0], dtype=np.float32)
    p = SF.PackedStations.from_host(pos, run, np.array([2, 0, 1]), "cpu")
    assert p.S == 3 and p.pos.dtype == torch.int32 and p.run.dtype == torch.float32
    assert p.pos.tolist() == pos.tolist() and p.run.tolist() == run.tolist() and p.offs.tolist() == [0, 2, 2, 3]
    assert SF.pack_stations(p, p, "cpu") == (p.pos, p.run, p.offs, 3)
    e = SF.PackedStations.from_host(np.zeros((0, 2)), np.zeros(0), np.array([0, 0]), "cpu")
    assert e.S == 0 and e.offs.tolist() == [0, 0, 0] and e.pos.shape == (1, 2)
    # identical to packing the list-of-tensors form
    lists_p = [torch.tensor([[3, 4], [5, 6]]), torch.zeros(0, 2, dtype=torch.long), torch.tensor([[7, 8]])]
    lists_r = [torch.tensor([0.5, -1.25]), torch.zeros(0), torch.tensor([2.0])]
    q = SF.pack_stations(lists_p, lists_r, "cpu")
    assert torch.equal(q[0], p.pos) and torch.equal(q[1], p.run) and torch.equal(q[2], p.offs) and q[3] == 3
