"""3x3 conv forward (256 -> 64, 400x400) at several batch sizes: small batches stay in the Infinity Cache."""
import os, sys, torch, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("tcn-transformer-gan-hybrid-model-for-predicting-streamflow-in-ungauged-basins_amd")
ops = pkg.ops
torch.manual_seed(0)
H, W, C, K = 400, 400, 256, 64
for N in (1, 2, 8, 32):
    x = torch.randn(N, H, W, C, device="cuda").bfloat16()
    wp = (torch.randn(K, 9, C, device="cuda") * 0.05).bfloat16()
    out = torch.empty(N, H, W, K, device="cuda", dtype=torch.bfloat16)
    reps = 64 // N
    for i in range(2): ops.conv3x3(x, wp, out, K)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps): ops.conv3x3(x, wp, out, K)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("N", N, "ms %.3f" % ms, "TF/s %.0f" % (2 * N * H * W * C * K * 9 / ms / 1e9), flush=True)
