import os, sys, torch, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("tcn-transformer-gan-hybrid-model-for-predicting-streamflow-in-ungauged-basins_amd")
ops = pkg.ops
torch.manual_seed(0)
N,H,W,C,K = 32,200,800,256,64
x = torch.randn(N,H,W,C, device="cuda").bfloat16()
dy = torch.randn(N,H,W,K, device="cuda").bfloat16()
dw = torch.zeros(K, C, 3, 3, device="cuda")
for i in range(3): ops.wgrad3x3(x, dy, dw)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(10): ops.wgrad3x3(x, dy, dw)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)/10
print("dbg", os.environ.get("SF_WGRAD_DBG"), "ms", ms, "TF/s", 2*N*H*W*C*K*9/ms/1e9)
