"""One 3x3 conv launch set (decoder conv1 forward shape) for counter passes."""
import os, sys, torch, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("tcn-transformer-gan-hybrid-model-for-predicting-streamflow-in-ungauged-basins_amd")
ops = pkg.ops
torch.manual_seed(0)
N, H, W, C, K = 32, 400, 400, int(os.environ.get("PC", 256)), int(os.environ.get("PK", 64))
x = torch.randn(N, H, W, C, device="cuda").bfloat16()
wp = (torch.randn(K, 9, C, device="cuda") * 0.05).bfloat16()
out = torch.empty(N, H, W, K, device="cuda", dtype=torch.bfloat16)
for i in range(3): ops.conv3x3(x, wp, out, K)
torch.cuda.synchronize()
