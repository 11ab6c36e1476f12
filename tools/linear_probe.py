"""The two 262144-wide Linear layers at the benchmark shape: time of forward and backward, and the
bytes each must move."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("tcn-transformer-gan-hybrid-model-for-predicting-streamflow-in-ungauged-basins_amd")
SF = importlib.import_module(pkg.__name__ + ".functional")
ops = pkg.ops
torch.manual_seed(0)


def timed(name, fn, nbytes, n=10):
    for _ in range(3):
        fn()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    us = 1e3 * a.elapsed_time(e) / n
    print(f"{name}: {us:.1f} us, {nbytes / us / 1e6:.2f} TB/s of {nbytes / 1e6:.0f} MB", flush=True)


for (M, K, N) in ((160, 262144, 128), (32, 128, 262144)):
    lin = torch.nn.Linear(K, N).cuda()
    sh = lin.weight.detach().to(torch.bfloat16)
    x = torch.randn(M, K, device="cuda", requires_grad=True)
    dy = torch.randn(M, N, device="cuda")
    y = SF.LinearFn.apply(x, lin.weight, lin.bias, torch.bfloat16, sh)
    fwd_bytes = M * K * 4 + M * K * 2 + N * K * 2 + M * N * 4          # x f32 read + bf16 write/read folded, W bf16, y
    timed(f"fwd  M={M} K={K} N={N}", lambda: SF.LinearFn.apply(x, lin.weight, lin.bias, torch.bfloat16, sh), fwd_bytes)

    def bwd():
        yy = SF.LinearFn.apply(x, lin.weight, lin.bias, torch.bfloat16, sh)
        yy.backward(dy)
        x.grad = None; lin.weight.grad = None; lin.bias.grad = None
    bwd_bytes = fwd_bytes + N * K * 2 + M * K * 4 + M * K * 2 + N * K * 4
    timed(f"fwd+bwd", bwd, bwd_bytes)
