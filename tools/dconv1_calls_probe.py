"""Prints the operand shapes of every first-layer PatchGAN weight-gradient call of a few bench steps."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sf = importlib.import_module("tcn-transformer-gan-hybrid-model-for-predicting-streamflow-in-ungauged-basins_amd")
ops = sf.ops
orig = ops.dconv1_bwd_weight


def spy(x, dz):
    print("dconv1_bwd_weight", tuple(x.shape), x.dtype, x.is_contiguous(), tuple(dz.shape), dz.dtype, dz.is_contiguous(), flush=True)
    return orig(x, dz)


ops.dconv1_bwd_weight = spy
import bench
sys.argv = ["bench.py", "--steps", "3", "--warmup", "0", "--no-cpu-baseline"]
bench.main()
