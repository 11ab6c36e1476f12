#!/bin/bash
# Stride-2 4x4 weight gradient on the LDS-DMA scheme: tests per ring depth, then the probe with the old and the new kernel.
export TMPDIR=/tmp
TAG=${1:-r3g}
mkdir -p gpurun_out
for ring in 3 2 4; do
SF_WGRAD4_RING=$ring timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize_conv.py -m gpu -q -x --timeout 200 --timeout-method thread -p no:cacheprovider -k "wgrad4x4s2 or patchgan" > gpurun_out/${TAG}_tests_s2_$ring.log 2>&1
echo "pytest rc=$?" >> gpurun_out/${TAG}_tests_s2_$ring.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=|Error|error" gpurun_out/${TAG}_tests_s2_$ring.log | tail -n 15
grep -q "rc=0" gpurun_out/${TAG}_tests_s2_$ring.log || { tail -n 60 gpurun_out/${TAG}_tests_s2_$ring.log; exit 1; }
done
SF_WGRAD4_OLD=1 timeout -k 10 120 python tools/s2_probe.py > gpurun_out/${TAG}_probe_old.log 2>&1 || { echo "probe old failed"; tail -n 5 gpurun_out/${TAG}_probe_old.log; exit 1; }
echo old; cat gpurun_out/${TAG}_probe_old.log
for ring in 2 3 4; do
SF_WGRAD4_RING=$ring timeout -k 10 120 python tools/s2_probe.py > gpurun_out/${TAG}_probe_new$ring.log 2>&1 || { echo "probe new failed"; tail -n 5 gpurun_out/${TAG}_probe_new$ring.log; exit 1; }
echo ring $ring; cat gpurun_out/${TAG}_probe_new$ring.log
done
