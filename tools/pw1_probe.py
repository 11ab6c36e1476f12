import os, sys, torch, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("tcn-transformer-gan-hybrid-model-for-predicting-streamflow-in-ungauged-basins_amd")
ops = pkg.ops
x = torch.randn(32, 400, 400, 64, device="cuda").bfloat16()
dy = torch.randn(32, 1, 400, 400, device="cuda")
def t(f, n=10):
    for i in range(2): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("cap", os.environ.get("SF_PW1_CAP"), "pw1_bwd_weight us %.1f" % t(lambda: ops.conv_k1_bwd_weight(x, dy, 64, 1, 0)))
print("channel_sum us %.1f" % t(lambda: ops.channel_sum(x, 64)))
y = torch.randn(160, 200, 200, 128, device="cuda").bfloat16()
print("cap", os.environ.get("SF_CAP"), "channel_sum big us %.1f" % t(lambda: ops.channel_sum(y, 128)))
z = torch.randn(32, 100, 100, 128, device="cuda").bfloat16()
print("channel_sum small us %.1f" % t(lambda: ops.channel_sum(z, 128)))
