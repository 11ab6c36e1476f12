#!/bin/bash
# conv3 LDS-DMA kernels with counted store drain: conv tests, then probe A/B against the draining build
export TMPDIR=/tmp
TAG=${1:-r3l}
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize_conv.py tests/test_gpu_conv_s2.py -m gpu -q -x --timeout 300 --timeout-method thread -p no:cacheprovider -k "conv or decoder or encoder or generator or discriminator" > gpurun_out/${TAG}_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/${TAG}_tests.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=|Error|error" gpurun_out/${TAG}_tests.log | tail -n 15
grep -q "rc=0" gpurun_out/${TAG}_tests.log || { tail -n 60 gpurun_out/${TAG}_tests.log; exit 1; }
shift
for n in "$@"; do
  echo "== $n"
  SF_LIB=$PWD/tools/diag/libsfhip_$n.so timeout -k 10 120 python tools/res_probe.py 2>&1 | grep -v amdgpu.ids || { echo "probe $n failed"; exit 1; }
  SF_LIB=$PWD/tools/diag/libsfhip_$n.so timeout -k 10 120 python tools/wide_probe.py 160 100 128 256 2>&1 | grep -v amdgpu.ids || { echo "probe $n failed"; exit 1; }
done > gpurun_out/${TAG}_diag.txt 2>&1
cat gpurun_out/${TAG}_diag.txt
