#!/bin/bash
# completeness items: large encoder filters, bidirectional LSTM, worker-fed loader
export TMPDIR=/tmp
TAG=${1:-r3p}
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_trainer_api.py -m gpu -v -x --timeout 400 --timeout-method thread -p no:cacheprovider -k "large_filters or bidirectional or constructor_options or lstm or worker or train_writes" > gpurun_out/${TAG}_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/${TAG}_tests.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=|Error|error" gpurun_out/${TAG}_tests.log | tail -n 15
grep -q "rc=0" gpurun_out/${TAG}_tests.log || { tail -n 80 gpurun_out/${TAG}_tests.log; exit 1; }
