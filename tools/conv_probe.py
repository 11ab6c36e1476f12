"""Times sf_conv3x3 on the decoder conv1 shapes (forward 256->64, data gradient 64->256)."""
import os, sys, torch, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("tcn-transformer-gan-hybrid-model-for-predicting-streamflow-in-ungauged-basins_amd")
ops = pkg.ops
torch.manual_seed(0)
N, H, W = 32, 400, 400
for (C, K) in ((256, 64), (64, 256), (64, 64), (128, 256), (16, 128)):
    if (C, K) == (128, 256):
        N, H, W = 160, 100, 100
    if (C, K) == (16, 128):
        N, H, W = 160, 200, 200
    x = torch.randn(N, H, W, C, device="cuda").bfloat16()
    wp = (torch.randn(K, 9, C, device="cuda") * 0.05).bfloat16()
    out = torch.empty(N, H, W, K, device="cuda", dtype=torch.bfloat16)
    for i in range(2): ops.conv3x3(x, wp, out, K)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(5): ops.conv3x3(x, wp, out, K)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("dbg", os.environ.get("SF_C3_DBG"), "old" if os.environ.get("SF_CONV3_OLD") else "new", (N, H, W, C, K), "ms %.3f" % ms, "TF/s %.0f" % (2 * N * H * W * C * K * 9 / ms / 1e9), flush=True)
    del x, out
