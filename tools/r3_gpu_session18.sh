#!/bin/bash
# Encoder tests, then a kernel trace of a short bench run (per-step kernel table).
export TMPDIR=/tmp
TAG=${1:-r3t}
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize_configs.py tests/test_gpu_fullsize_temporal.py -q -x -k "enc or generator or config or transformer or temporal" --timeout 200 --timeout-method thread -p no:cacheprovider > gpurun_out/${TAG}_tests.log 2>&1 || { echo "tests failed"; tail -n 30 gpurun_out/${TAG}_tests.log; exit 1; }
tail -n 2 gpurun_out/${TAG}_tests.log
timeout -s KILL 240 rocprofv3 --kernel-trace -d gpurun_out/${TAG}_trace -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_trace.log 2>&1 || { echo "trace failed"; tail -n 5 gpurun_out/${TAG}_trace.log; exit 1; }
python tools/prof_steps.py $(ls -t gpurun_out/${TAG}_trace/*/*.db | head -n 1) 12 gpurun_out/${TAG}_kernel_stats.csv > gpurun_out/${TAG}_steps.txt 2>&1
head -n 3 gpurun_out/${TAG}_steps.txt
grep -E "transformer|enc1_gram_fold|nchw_to_nhwc_px4|dconv1_bwd_weight|nhwc_to_nchw" gpurun_out/${TAG}_kernel_stats.csv | cut -c1-60,100-200
rm -rf gpurun_out/${TAG}_trace
