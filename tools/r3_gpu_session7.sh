#!/bin/bash
# weight gradients with 128 x 64 blocks: tests (fixed slices and tile queue), then the probe at the benchmark shapes.
export TMPDIR=/tmp
TAG=${1:-r3h}
mkdir -p gpurun_out
timeout -k 10 500 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize_conv.py tests/test_gpu_dp.py -m gpu -q -x --timeout 300 --timeout-method thread -p no:cacheprovider -k "wgrad or patchgan or encoder or conv3 or stealing or ranks or dp" > gpurun_out/${TAG}_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/${TAG}_tests.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=|Error|error" gpurun_out/${TAG}_tests.log | tail -n 15
grep -q "rc=0" gpurun_out/${TAG}_tests.log || { tail -n 60 gpurun_out/${TAG}_tests.log; exit 1; }
for wide in 0 1 0 1; do
SF_WGRAD3_WIDE=$wide timeout -k 10 120 python tools/wgrad_probe.py 160,200,64,128 32,400,64,64 > gpurun_out/${TAG}_probe_$wide.log 2>&1 || { echo "probe failed"; tail -n 5 gpurun_out/${TAG}_probe_$wide.log; exit 1; }
echo wide $wide; grep -v amdgpu.ids gpurun_out/${TAG}_probe_$wide.log
done
