"""Times the PatchGAN stride-2 convolutions (forward, data gradient, weight gradient) at the
benchmark's three layer shapes: generic implicit GEMM against the LDS-DMA kernel."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sf = importlib.import_module("tcn-transformer-gan-hybrid-model-for-predicting-streamflow-in-ungauged-basins_amd")
fn = importlib.import_module(sf.__name__ + ".functional")
ops = sf.ops
DEV = "cuda"


def timeit(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for (IH, C, K) in ((200, 64, 128), (100, 128, 256), (50, 256, 512)):
    N, OH = 32, IH // 2
    x = torch.randn(N, IH, IH, C, device=DEV).to(torch.bfloat16)
    w = (torch.randn(K, C, 4, 4, device=DEV) / (4.0 * C ** 0.5))
    b = torch.randn(K, device=DEV)
    z = torch.empty(N, OH, OH, K, device=DEV, dtype=torch.bfloat16)
    dz = torch.randn(N, OH, OH, K, device=DEV).to(torch.bfloat16)
    dx = torch.empty_like(x)
    dw = torch.empty(K, C, 4, 4, device=DEV)
    fl = 2.0 * N * OH * OH * C * K * 16
    wp = ops.conv4x4s2_weights(w, False)
    wT = ops.conv4x4s2_weights(w, True)
    wg = ops.pack_conv_weight(w, torch.bfloat16, list(range(16)), False, K, C)
    geom = fn._s2_geom(N, IH, IH, C, OH, OH)
    t_new = timeit(lambda: ops.conv4x4s2(x, wp, z, K, bias=b))
    t_old = timeit(lambda: ops.conv_fprop(x, wg, z, geom, K, bias=b))
    t_dnew = timeit(lambda: ops.conv4x4s2(dz, wT, dx, C, dgrad=True))

    def old_dgrad():
        problems = []
        for (py, px, taps, rs, mh, mw) in fn._s2_dgrad_classes(IH, IH):
            wc = ops.pack_conv_weight(w, torch.bfloat16, rs, True, C, K)
            g = ops.conv_geom(N, OH, OH, K, mh, mw, 1, 1, IH, IH, taps, osh=2, osw=2, ooy=py, oox=px)
            problems.append((dz, wc, dx, g, C, None, 0))
        ops.conv_fprop_batch(problems)

    t_dold = timeit(old_dgrad)
    t_wg = timeit(lambda: ops.wgrad4x4s2(x, dz, dw))
    t_dmask = timeit(lambda: ops.conv4x4s2(dz, wT, dx, C, dgrad=True, aux=x, aux_mode=2))
    print(f"{IH}x{IH} {C}->{K}: fwd {t_new * 1e3:.0f} us ({fl / t_new / 1e9:.0f} TF/s) vs generic {t_old * 1e3:.0f} us; "
          f"dgrad {t_dnew * 1e3:.0f} us ({fl / t_dnew / 1e9:.0f} TF/s) vs generic {t_dold * 1e3:.0f} us; "
          f"wgrad {t_wg * 1e3:.0f} us ({fl / t_wg / 1e9:.0f} TF/s); masked dgrad {t_dmask * 1e3:.0f} us", flush=True)
