"""Device-resident dataset kernels on a year of full-size frames (365 dates x 10 channels x
200 x 200, 584 MB; src/dataset.py:191-317 statistics, :425-507 per-sample transform, :379-424
window assembly): sf_dataset_plane_stats against float64 torch reductions, sf_dataset_transform
against the same arithmetic spelled out in torch (bit-exact: one float64 subtraction and division,
rounded to float32 once), sf_gather_frames against indexing (bit-exact), with masks on some dates
and NaN / Inf pixels."""
import pytest
import torch

from conftest import load_pkg

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def sf():
    return load_pkg()


def test_year_of_frames(sf):
    ops = sf.ops
    D, C, H, W = 365, 10, 200, 200
    g = torch.Generator(device=DEV).manual_seed(20)
    frames = torch.rand(D, C, H, W, device=DEV, generator=g) * 300.0
    frames[:, 0][frames[:, 0] < 30.0] *= 0.001                    # channel 0 values below the flow threshold
    frames[7, 3, 5, 9] = float("nan")
    frames[11, 0, 0, 0] = float("nan")
    masks = (torch.rand(2, H * W, device=DEV, generator=g) > 0.3).float()
    mask_row = torch.full((D,), -1, device=DEV, dtype=torch.int32)
    mask_row[5::7] = 0
    mask_row[6::11] = 1
    stats, count, bad0 = ops.dataset_plane_stats(frames, masks, mask_row)
    x64 = torch.nan_to_num(frames, nan=0.0).double().view(D, C, H * W)
    sel = torch.ones(D, H * W, device=DEV, dtype=torch.bool)
    for r in (0, 1):
        sel[mask_row == r] = masks[r] > 0.5
    selc = sel.view(D, 1, H * W)
    assert torch.equal(count, sel.sum(1))
    assert int(bad0[11]) == 1 and int(bad0.sum()) == 1              # non-finite pixels of channel 0 only
    ref_s = (x64 * selc).sum(2)
    ref_ss = ((x64.float() * x64.float()).double() * selc).sum(2)   # squares in float32, as `vals ** 2` on a float32 array
    assert float((stats[..., 0] - ref_s).abs().max() / ref_s.abs().max()) <= 1e-12
    assert float((stats[..., 1] - ref_ss).abs().max() / ref_ss.abs().max()) <= 1e-12
    big = torch.finfo(torch.float64).max
    assert torch.equal(stats[..., 2], torch.where(selc, x64, torch.full_like(x64, big)).min(2).values)
    assert torch.equal(stats[..., 3], torch.where(selc, x64, torch.full_like(x64, -big)).max(2).values)
    # transform
    a = (ref_s.sum(0) / count.sum()).contiguous()
    b = (ref_ss.sum(0) / count.sum() - a * a).sqrt().contiguous()
    work = frames.clone()
    target = torch.empty(D, 1, H, W, device=DEV)
    ops.dataset_transform(work, target, masks, mask_row, a, b, 0.1)
    v = frames.view(D, C, H * W).clone()
    for r in (0, 1):
        v[mask_row == r] *= masks[r]
    t = v[:, 0].clone()
    v = torch.where(torch.isfinite(v), v, torch.zeros_like(v))
    v[:, 0] = torch.where(v[:, 0].abs() < 0.1, torch.zeros_like(v[:, 0]), v[:, 0])
    ref = ((v.double() - a.view(1, C, 1)) / b.view(1, C, 1)).float()
    assert torch.equal(work.view(D, C, H * W), ref)
    t = torch.where(t.abs() < 0.1, torch.zeros_like(t), t)          # NaN stays NaN in the target
    reft = ((t.double() - a[0]) / b[0]).float()
    assert torch.equal(torch.nan_to_num(target.view(D, H * W), nan=-7.0), torch.nan_to_num(reft, nan=-7.0))
    # windows of five consecutive dates for 32 shuffled start dates, and their targets
    starts = torch.randperm(D - 5, device=DEV, generator=g)[:32]
    rows = (starts.view(-1, 1) + torch.arange(5, device=DEV)).reshape(-1)
    out = torch.empty(32, 5, C, H, W, device=DEV)
    ops.gather_frames(work, C * H * W, D, rows, C * H * W, out)
    assert torch.equal(out, work[rows].view(32, 5, C, H, W))
    tout = torch.empty(32, 1, H, W, device=DEV)
    ops.gather_frames(target, H * W, D, starts + 5, H * W, tout)
    assert torch.equal(torch.nan_to_num(tout, nan=-7.0), torch.nan_to_num(target[starts + 5], nan=-7.0))


def test_more_dates_and_rows_than_a_grid_dimension_holds(sf):
    """70 000 dates / gathered rows (a grid's y and z dimensions hold 65 535 blocks: the wrappers
    launch in chunks): statistics against float64 reductions, transform and gather bit-exact."""
    ops = sf.ops
    D, C, H, W = 70000, 2, 2, 4
    g = torch.Generator().manual_seed(5)
    frames = torch.randn(D, C, H, W, generator=g).to(DEV)
    stats, count, bad0 = ops.dataset_plane_stats(frames, None, None)
    f64 = frames.double().view(D, C, -1)
    assert torch.allclose(stats[:, :, 0], f64.sum(-1), rtol=1e-12, atol=1e-12)
    assert torch.equal(stats[:, :, 2], f64.min(-1).values) and torch.equal(stats[:, :, 3], f64.max(-1).values)
    assert int(count.min()) == H * W and int(count.max()) == H * W and int(bad0.sum()) == 0
    rows = torch.randperm(D, generator=g).to(DEV)
    out = torch.empty(D, C * H * W, device=DEV)
    ops.gather_frames(frames, C * H * W, D, rows, C * H * W, out)
    assert torch.equal(out, frames.view(D, -1)[rows])
    a = torch.tensor([0.25, -1.0], dtype=torch.float64, device=DEV)
    b = torch.tensor([2.0, 0.5], dtype=torch.float64, device=DEV)
    want = ((frames.double() - a.view(1, C, 1, 1)) / b.view(1, C, 1, 1)).float()
    target = torch.empty(D, 1, H, W, device=DEV)
    ops.dataset_transform(frames, target, None, None, a, b, 0.0)
    assert torch.equal(frames, want) and torch.equal(target[:, 0], want[:, 0])
