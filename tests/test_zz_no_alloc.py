"""sf_image_station_loss_static (the entry point for callers that cannot allocate: its partial
sums live in a static device array of the library) against sf_image_station_loss on the same
inputs, through the C ABI: two calls in a row, the second on a smaller input, so that what the
first call left in the shared array must not leak into the second result.  The tolerance is the
forward / loss bar of the f32 path, 1e-5 (the losses are bit-reproducible; the station gradient
is scattered with float atomics, so overlapping windows may differ in the last bit)."""
import numpy as np
import pytest
import torch

from conftest import close, load_pkg

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _case(sf, B, H, W, S, seed, upsample=True):
    g = torch.Generator().manual_seed(seed)
    pred = torch.randn(B, 1, H, W, generator=g)
    tgt = torch.randn(B, 1, H // 2, W // 2, generator=g) if upsample else torch.randn(B, 1, H, W, generator=g)
    tgt[:, :, :3, :5] = 0.0
    pos, run = [], []
    for b in range(B):
        n = 0 if (b == 1 and B > 2) else max(S - b % 3, 0)
        p = torch.stack([torch.randint(0, W, (n,), generator=g), torch.randint(0, H, (n,), generator=g)], 1)
        if n >= 3:
            p[0] = torch.tensor([0, 0]); p[1] = torch.tensor([W - 1, H - 1]); p[2] = p[0]
        pos.append(p.to(torch.int64))
        run.append(torch.randn(n, generator=g))
    dpos, drun, doffs, Stot = sf.functional.pack_stations(pos, run, DEV)
    return pred.to(DEV), tgt.to(DEV), dpos.contiguous(), drun.contiguous(), doffs.contiguous(), Stot


def _both(sf, case):
    pred, tgt, dpos, drun, doffs, S = case
    args = (0, 10.0, 1e-6, 5, 0, 1.0, 0.5)
    ref_losses, ref_dpred = sf.ops.image_station_loss(pred, tgt, dpos, drun, doffs, S, *args)
    losses = torch.full((4,), float("nan"), device=DEV)
    dpred = torch.full_like(pred, float("nan"))
    sf.ops.image_station_loss_static(pred, tgt, dpos, drun, doffs, S, *args, losses, dpred)
    torch.cuda.synchronize()
    return ref_losses.cpu().numpy(), ref_dpred.cpu().numpy(), losses.cpu().numpy(), dpred.cpu().numpy()


def test_static_scratch_loss_matches_workspace_loss_twice_in_a_row():
    sf = load_pkg()
    big = _case(sf, 8, 400, 400, 64, seed=11)
    small = _case(sf, 3, 64, 96, 7, seed=12)
    same = _case(sf, 2, 40, 40, 0, seed=13, upsample=False)      # no stations, same-size target
    for case in (big, small, same):
        rl, rd, l, d = _both(sf, case)
        assert np.isfinite(l).all() and np.isfinite(d).all()
        np.testing.assert_array_equal(l, rl)          # partial sums folded in a fixed order: bit-identical
        close(d, rd, 1e-5)


def test_static_scratch_loss_rejects_sizes_beyond_its_limit():
    sf = load_pkg()
    pred, tgt, dpos, drun, doffs, S = _case(sf, 1, 8, 8, 2, seed=14)
    losses = torch.zeros(4, device=DEV)
    lib = sf.ops.lib()
    import ctypes
    vp = ctypes.c_void_p
    f = ctypes.c_float
    # 2^15 x 2^15 pixels claimed (beyond SF_LOSS_STATIC_MAX_PIXELS): refused before anything is launched
    rc = lib.sf_image_station_loss_static(vp(pred.data_ptr()), vp(tgt.data_ptr()), vp(dpos.data_ptr()), vp(drun.data_ptr()),
                                          vp(doffs.data_ptr()), S, 1, 1 << 15, 1 << 15, 4, 4, 0, f(10.0), f(1e-6), 5, 0,
                                          f(1.0), f(0.5), vp(losses.data_ptr()), vp(pred.data_ptr()), vp(0))
    assert rc == -22
    assert b"limit" in lib.sf_last_error()
