"""PatchGAN head (4x4 conv to one channel) at the benchmark shape: time per launch of the forward,
data-gradient and weight-gradient calls."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("tcn-transformer-gan-hybrid-model-for-predicting-streamflow-in-ungauged-basins_amd")
ops = pkg.ops
torch.manual_seed(0)
N, H, C = 32, 25, 512
x = torch.randn(N, H, H, C, device="cuda").bfloat16()
w = torch.randn(1, C, 4, 4, device="cuda") * 0.02
b = torch.zeros(1, device="cuda")
y = ops.conv_k1_fwd(x, w, b, 4, 1)
dy = torch.randn_like(y)


def timed(name, fn, n=20):
    for _ in range(3):
        fn()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    print(f"{name}: {1e3 * a.elapsed_time(e) / n:.1f} us", flush=True)


timed("fwd", lambda: ops.conv_k1_fwd(x, w, b, 4, 1))
timed("bwd_data", lambda: ops.conv_k1_bwd_data(dy, w, x, 4, 1, relu_mask=False))
timed("bwd_weight", lambda: ops.conv_k1_bwd_weight(x, dy, C, 4, 1))
ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w, b, padding=1)
xg = x.float().permute(0, 3, 1, 2).requires_grad_(True)
torch.nn.functional.conv2d(xg, w, b, padding=1).backward(dy.reshape(ref.shape))
dx = ops.conv_k1_bwd_data(dy, w, x, 4, 1, relu_mask=False)
print("bwd_data err", float((dx.float().permute(0, 3, 1, 2) - xg.grad).abs().max() / xg.grad.abs().max()))
dxm = ops.conv_k1_bwd_data(dy, w, x, 4, 1, relu_mask=True)
want = xg.grad * (x.float().permute(0, 3, 1, 2) > 0)
print("bwd_data masked err", float((dxm.float().permute(0, 3, 1, 2) - want).abs().max() / want.abs().max()))
print("fwd err", float((y.reshape(ref.shape) - ref).abs().max() / ref.abs().max()))
