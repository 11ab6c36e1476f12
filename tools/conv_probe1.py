"""One 3x3 conv shape (env PC / PK / PN / PHW; defaults: decoder conv1 forward) timed with events, for counter passes."""
import os, sys, torch, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("tcn-transformer-gan-hybrid-model-for-predicting-streamflow-in-ungauged-basins_amd")
ops = pkg.ops
torch.manual_seed(0)
N, H, C, K = int(os.environ.get("PN", 32)), int(os.environ.get("PHW", 400)), int(os.environ.get("PC", 256)), int(os.environ.get("PK", 64))
W = H
x = torch.randn(N, H, W, C, device="cuda").bfloat16()
wp = (torch.randn(K, 9, C, device="cuda") * 0.05).bfloat16()
out = torch.empty(N, H, W, K, device="cuda", dtype=torch.bfloat16)
for i in range(3): ops.conv3x3(x, wp, out, K)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(5): ops.conv3x3(x, wp, out, K)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print("C", C, "K", K, "N", N, "HW", H, "ms %.3f" % ms, "TF/s %.0f" % (2 * N * H * W * C * K * 9 / ms / 1e9), flush=True)
