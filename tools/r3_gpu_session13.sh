#!/bin/bash
# bench twice, kernel trace of the tree, full GPU test suite.
export TMPDIR=/tmp
TAG=${1:-r3n}
mkdir -p gpurun_out
for i in 1 2; do
  timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/${TAG}_bench_new$i.log 2>/dev/null || { echo "bench new failed"; exit 1; }
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob("gpurun_out/${TAG}_bench_*.log")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), d["ms_per_step"], d["timing"]["ms_per_step_hip_events"], d["final_loss"], d["roofline"]["frac"])
PY
timeout -s KILL 240 rocprofv3 --kernel-trace -d gpurun_out/${TAG}_trace -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_trace.log 2>&1 || { echo "trace failed"; tail -n 5 gpurun_out/${TAG}_trace.log; exit 1; }
python tools/prof_steps.py $(ls -t gpurun_out/${TAG}_trace/*/*.db | head -n 1) 36 gpurun_out/${TAG}_kernel_stats.csv > gpurun_out/${TAG}_steps.txt 2>&1
head -n 3 gpurun_out/${TAG}_steps.txt
rm -rf gpurun_out/${TAG}_trace
timeout -k 10 1000 python -u -m pytest tests -m gpu -q --timeout 300 --timeout-method thread -p no:cacheprovider > gpurun_out/${TAG}_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/${TAG}_tests.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=" gpurun_out/${TAG}_tests.log | tail -n 15
