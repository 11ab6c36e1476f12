"""sf_wgrad3x3 at the benchmark shapes: time per launch with fixed slices, with stealing allowed, and
with every second workgroup leaving its list to the others."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
sf = importlib.import_module(bench.PKG)
ops = sf.ops
for (N, H, C, K) in [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or ((160, 100, 128, 256), (32, 400, 64, 64), (32, 200, 128, 128)):
    x = torch.randn(N, H, H, C, device="cuda").to(torch.bfloat16)
    dz = torch.randn(N, H, H, K, device="cuda").to(torch.bfloat16)
    dw = torch.empty(K, C, 3, 3, device="cuda")
    out = []
    for mode in (0, 1, 2, 0, 1):
        ops.set_wgrad_scheduling(mode)
        for _ in range(3):
            ops.wgrad3x3(x, dz, dw)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            ops.wgrad3x3(x, dz, dw)
        b.record()
        torch.cuda.synchronize()
        out.append(f"mode {mode}: {a.elapsed_time(b) / 20:.4f} ms")
    ops.set_wgrad_scheduling(0)
    print((N, H, C, K), "  ".join(out), flush=True)
