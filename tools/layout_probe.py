"""Times the few-channel NCHW f32 -> NHWC bf16 layout pass at the bench shape and checks it against
torch (exact: the pass only rounds and moves).  Usage: python tools/layout_probe.py"""
import importlib
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sf = importlib.import_module("tcn-transformer-gan-hybrid-model-for-predicting-streamflow-in-ungauged-basins_amd")
ops = sf.ops


def check(N, C, H, W, cpad, with_flag):
    g = torch.Generator(device="cuda").manual_seed(N * 131 + C)
    x = torch.randn((N, C, H, W), device="cuda", generator=g)
    flag = torch.zeros(1, device="cuda", dtype=torch.int32) if with_flag else None
    y = ops.nchw_to_nhwc(x, torch.bfloat16, cpad, flag)
    ref = torch.zeros((N, H, W, cpad), device="cuda", dtype=torch.bfloat16)
    ref[..., :C] = x.permute(0, 2, 3, 1).to(torch.bfloat16)
    assert torch.equal(y, ref), (N, C, H, W, cpad)
    if with_flag:
        assert int(flag.item()) == 0
        x[N - 1, C - 1, H - 1, W - 1] = float("nan")
        ops.nchw_to_nhwc(x, torch.bfloat16, cpad, flag)
        assert int(flag.item()) == 1


for shape in [(160, 10, 200, 200, 16, True), (3, 10, 6, 6, 16, False), (1, 3, 2, 2, 8, True), (5, 7, 10, 6, 8, False),
              (2, 13, 17, 4, 16, True), (7, 1, 3, 4, 8, False)]:
    check(*shape)
print("layout pass matches torch on all shapes")
x = torch.randn((160, 10, 200, 200), device="cuda")
flag = torch.zeros(1, device="cuda", dtype=torch.int32)
for _ in range(5):
    ops.nchw_to_nhwc(x, torch.bfloat16, 16, flag)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    ops.nchw_to_nhwc(x, torch.bfloat16, 16, flag)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 50 * 1e3
mb = (x.numel() * 4 + 160 * 200 * 200 * 16 * 2) / 1e6
print(f"[160,10,200,200] f32 -> NHWC16 bf16: {us:.1f} us, {mb / us:.2f} TB/s algorithmic")
