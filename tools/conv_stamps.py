"""Stage / barrier / epilogue time inside the persistent 3x3 conv kernel, from the diagnostic
build with shader-clock stamps (tools/diag/libsfhip_stamps.so: sf_conv3x3.hip compiled with
-DSF_CONV3_STAMPS).  env PC / PK / PN / PHW choose the shape."""
import ctypes
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("tcn-transformer-gan-hybrid-model-for-predicting-streamflow-in-ungauged-basins_amd")
ops = pkg.ops
ops.LIB_PATH = os.path.join(ROOT, "tools", "diag", os.environ.get("SF_DIAG_LIB", "libsfhip_stamps.so"))
ops._lib = None
N, H, C, K = int(os.environ.get("PN", 32)), int(os.environ.get("PHW", 400)), int(os.environ.get("PC", 64)), int(os.environ.get("PK", 64))
torch.manual_seed(0)
x = torch.randn(N, H, H, C, device="cuda").bfloat16()
wp = (torch.randn(K, 9, C, device="cuda") * 0.05).bfloat16()
ws = ops.conv3x3_stage_weights(wp, C, K)
out = torch.empty(N, H, H, K, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    ops.conv3x3(x, ws, out, K)
torch.cuda.synchronize()
STEPS = 48
buf = np.zeros(256 * STEPS * 8, dtype=np.uint64)
rc = ops.lib().sf_conv3x3_read_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.size)
assert rc == 0
s = buf.reshape(256, STEPS, 8).astype(np.int64)
bar = s[:, :, 1] - s[:, :, 0]
stage = s[:, :, 2] - s[:, :, 1]
epi = s[:, :, 3] - s[:, :, 2]
tot = s[:, 1:, 0] - s[:, :-1, 0]
sl = slice(4, STEPS - 2)
print(f"C {C} K {K}: per pipeline step (shader-clock ticks, median over workgroups and steps {sl.start}..{sl.stop}):")
print("  barrier wait  ", np.median(bar[:, sl]), " mean", bar[:, sl].mean())
print("  stage (MFMA)  ", np.median(stage[:, sl]), " mean", stage[:, sl].mean())
print("  after stage   ", np.median(epi[:, sl]), " mean", epi[:, sl].mean(), " (epilogue steps:", np.median(epi[:, sl][epi[:, sl] > 500]) if (epi[:, sl] > 500).any() else 0, ")")
print("  step total    ", np.median(tot[:, sl]), " mean", tot[:, sl].mean())
ep = s[:, sl, :][(s[:, sl, 4] > 0) & (s[:, sl, 5] > 0) & (epi[:, sl] > 500)]
if len(ep):
    print("  epilogue steps: barrier before", np.median(ep[:, 4] - ep[:, 2]), " epilogue body", np.median(ep[:, 5] - ep[:, 4]),
          " rest", np.median(ep[:, 3] - ep[:, 5]), " whole", np.median(ep[:, 3] - ep[:, 2]))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.conv3x3(x, ws, out, K)
e1.record()
torch.cuda.synchronize()
print(f"  kernel {e0.elapsed_time(e1) / 10:.4f} ms")
