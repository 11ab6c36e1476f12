"""Decoder per-source-pixel GEMMs: conv_fprop (gather path) vs linear_gemm (plain) timings."""
import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
sf = importlib.import_module(bench.PKG)
ops = sf.ops
B, h, w, C, K9 = 32, 32, 32, 256, 576
dt = torch.bfloat16
src = torch.randn(B, h, w, C, device="cuda").to(dt)
w9 = torch.randn(K9, C, 1, 1, device="cuda")
wp = ops.pack_conv_weight(w9, dt, [0], False, K9, C)
wc = ops.cast(w9.view(K9, C).contiguous(), dt)
geom = ops.conv_geom(B, h, w, C, h, w, 1, 1, h, w, [(0, 0)], tap_rs=[0])
y1 = torch.empty(B, h, w, K9, device="cuda")
y2 = torch.empty(B * h * w, K9, device="cuda")
g = torch.randn(B, h, w, K9, device="cuda").to(dt)
wT = ops.pack_conv_weight(w9, dt, [0], True, C, K9)
wTc = ops.cast(w9.view(K9, C).t().contiguous(), dt)
d1 = torch.empty(B, h, w, C, device="cuda")
d2 = torch.empty(B * h * w, C, device="cuda")
geom2 = ops.conv_geom(B, h, w, K9, h, w, 1, 1, h, w, [(0, 0)], tap_rs=[0])
dw1 = torch.zeros(K9, C, 1, 1, device="cuda")
dw2 = torch.zeros(K9, C, device="cuda")
wg = ops.conv_geom(B, h, w, C, h, w, 1, 1, h, w, [(0, 0)], tap_rs=[0], cin_real=C, RS=1)
M = B * h * w
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("Y   conv_fprop  %.1f us" % t(lambda: ops.conv_fprop(src, wp, y1, geom, K9)))
print("Y   linear_gemm %.1f us" % t(lambda: ops.linear_gemm(wc, src.view(M, C), y2, K9, M, C, C, C, K9, False, False, ops.OUT_JI, 1)))
print("max diff", float((y1.view(M, K9) - y2).abs().max()))
print("dx  conv_fprop  %.1f us" % t(lambda: ops.conv_fprop(g, wT, d1, geom2, C)))
print("dx  linear_gemm %.1f us" % t(lambda: ops.linear_gemm(wTc, g.view(M, K9), d2, C, M, K9, K9, K9, C, False, False, ops.OUT_JI, 1)))
print("max diff", float((d1.view(M, C) - d2).abs().max()))
def wg1():
    dw1.zero_(); ops.conv_wgrad(src, g, dw1, wg, K9)
for ns in (1, 8, 32, 64):
    def wg2():
        dw2.zero_(); ops.linear_gemm(g.view(M, K9), src.view(M, C), dw2, K9, C, M, K9, C, C, True, True, ops.OUT_IJ, ns)
    print("dW  linear_gemm nsplit", ns, "%.1f us" % t(wg2))
print("dW  conv_wgrad  %.1f us" % t(wg1))
print("max diff", float((dw1.view(K9, C) - dw2).abs().max() / dw1.abs().max()))
