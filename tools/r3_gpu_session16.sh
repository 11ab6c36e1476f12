#!/bin/bash
# wide weight-gradient kernel: tests, then the probes with two builds of the library in turn
export TMPDIR=/tmp
TAG=${1:-r3x}
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize_conv.py -m gpu -q -x --timeout 300 --timeout-method thread -p no:cacheprovider -k "wgrad or patchgan or stealing or encoder or discriminator" > gpurun_out/${TAG}_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/${TAG}_tests.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=|Error|error" gpurun_out/${TAG}_tests.log | tail -n 15
grep -q "rc=0" gpurun_out/${TAG}_tests.log || { tail -n 60 gpurun_out/${TAG}_tests.log; exit 1; }
shift
for n in "$@"; do
  echo "== $n"
  SF_LIB=$PWD/tools/diag/libsfhip_$n.so timeout -k 10 120 python tools/wgrad_probe.py 32,400,64,64 2>&1 | grep -v amdgpu.ids | cut -c1-120 || { echo "probe $n failed"; exit 1; }
  SF_LIB=$PWD/tools/diag/libsfhip_$n.so timeout -k 10 120 python tools/s2_probe.py 2>&1 | grep -o "^[0-9x]* [0-9>-]*\|wgrad.*" | paste - - || { echo "probe $n failed"; exit 1; }
done > gpurun_out/${TAG}_diag.txt 2>&1
cat gpurun_out/${TAG}_diag.txt
