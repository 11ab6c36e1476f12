"""Per-wave stage times inside the persistent 3x3 conv kernel, from the diagnostic build with
shader-clock stamps of every wave (tools/build_diag.sh -DSF_CONV3_STAMPS_ALLWAVES [...]).  Prints, per
wave index, the median time from the barrier's release to the end of the stage's MFMAs and the wait
at the next barrier, and which wave is last.  env PC / PK / PN / PHW choose the shape."""
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
N, H, C, K = int(os.environ.get("PN", 160)), int(os.environ.get("PHW", 100)), int(os.environ.get("PC", 128)), int(os.environ.get("PK", 256))
torch.manual_seed(0)
x = torch.randn(N, H, H, C, device="cuda").bfloat16()
wp = (torch.randn(K, 9, C, device="cuda") * 0.05).bfloat16()
ws = ops.conv3x3_stage_weights(wp, C, K)
out = torch.empty(N, H, H, K, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    ops.conv3x3(x, ws, out, K)
torch.cuda.synchronize()
STEPS = 48
buf = np.zeros(256 * STEPS * 8 * 8, dtype=np.uint64)
assert ops.lib().sf_conv3x3_read_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.size) == 0
s = buf.reshape(256, STEPS, 8, 8).astype(np.int64)       # [workgroup][step][slot][wave]
sl = slice(4, STEPS - 2)
rel = s[:, sl, 1, :]                 # after the barrier
end = s[:, sl, 2, :]                 # after the stage's MFMAs
nxt = s[:, 5:STEPS - 1, 0, :]        # arrival at the next barrier
nrel = s[:, 5:STEPS - 1, 1, :]
first = rel.min(axis=2, keepdims=True)
print("wave:                      ", "  ".join(f"{w:6d}" for w in range(8)))
print("release after first wave:  ", "  ".join(f"{v:6.0f}" for v in np.median(rel - first, axis=(0, 1))))
print("stage end after release:   ", "  ".join(f"{v:6.0f}" for v in np.median(end - first, axis=(0, 1))))
print("arrival at next barrier:   ", "  ".join(f"{v:6.0f}" for v in np.median(nxt - first, axis=(0, 1))))
print("next release:              ", "  ".join(f"{v:6.0f}" for v in np.median(nrel - first, axis=(0, 1))))
last = (nxt - first).argmax(axis=2)
print("last wave to arrive (share):", "  ".join(f"{(last == w).mean():6.2f}" for w in range(8)))
if hasattr(ops.lib(), "sf_conv3x3_read_pos_stamps"):
    pb = np.zeros(256 * STEPS * 8 * 20, dtype=np.uint64)
    assert ops.lib().sf_conv3x3_read_pos_stamps(pb.ctypes.data_as(ctypes.c_void_p), pb.size) == 0
    ps = pb.reshape(256, STEPS, 8, 20).astype(np.int64)[:, sl]           # [workgroup][step][wave][position]
    npos = int((ps[0, 0, 0] > 0).sum())
    print(f"end of tap position 0..{npos - 1} after the release, per wave (median ticks):")
    for w in range(8):
        d = ps[:, :, w, :npos] - first[:, :, 0, None]
        print(f"  wave {w}: ", "  ".join(f"{v:5.0f}" for v in np.median(d, axis=(0, 1))))
