#!/bin/bash
# conv3 LDS-DMA kernels: timing-only diagnostic builds on the probe shapes (resident 64->64 and the encoder's 128->256)
export TMPDIR=/tmp
TAG=${1:-r3k}
mkdir -p gpurun_out
for n in "$@"; do
  [ "$n" = "$TAG" ] && continue
  echo "== $n"
  SF_LIB=$PWD/tools/diag/libsfhip_$n.so timeout -k 10 120 python tools/res_probe.py 2>&1 | grep -v amdgpu.ids || { echo "probe $n failed"; exit 1; }
  SF_LIB=$PWD/tools/diag/libsfhip_$n.so timeout -k 10 120 python tools/wide_probe.py 160 100 128 256 2>&1 | grep -v amdgpu.ids || { echo "probe $n failed"; exit 1; }
done > gpurun_out/${TAG}_diag.txt 2>&1
cat gpurun_out/${TAG}_diag.txt
