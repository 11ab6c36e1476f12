#!/bin/bash
# Batch-size sweep of the default bench (one GPU) and the smoke entry point.  Each GPU step has its
# own time limit; the chain stops at the first failure.
export TMPDIR=/tmp
TAG=${1:-r3sweep}
mkdir -p gpurun_out
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1 || { echo "smoke failed"; tail -n 20 gpurun_out/${TAG}_smoke.log; exit 1; }
tail -n 3 gpurun_out/${TAG}_smoke.log
for B in 8 16 32 64 128; do
  timeout -k 10 200 python bench.py --batch $B --no-cpu-baseline > gpurun_out/${TAG}_b${B}.log 2> gpurun_out/${TAG}_b${B}.err || { echo "bench B=$B failed"; tail -n 20 gpurun_out/${TAG}_b${B}.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/${TAG}_b${B}.log").read().strip().splitlines()[-1])
print("B=%d: %.1f samples/s, %.3f ms/step, wide %.4f ms frac %.3f" % ($B, d["value"], d["ms_per_step"], d["roofline"]["ms_per_launch"], d["roofline"]["frac"]))
PY
done
