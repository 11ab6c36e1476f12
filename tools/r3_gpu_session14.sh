#!/bin/bash
# data-parallel tests (incl. sharded big parameters) and the bench rehearsals
export TMPDIR=/tmp
TAG=${1:-r3o}
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_gpu_dp.py tests/test_gpu_bench_line.py -m gpu -q -x --timeout 400 --timeout-method thread -p no:cacheprovider > gpurun_out/${TAG}_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/${TAG}_tests.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=|Error|error" gpurun_out/${TAG}_tests.log | tail -n 15
grep -q "rc=0" gpurun_out/${TAG}_tests.log || { tail -n 80 gpurun_out/${TAG}_tests.log; exit 1; }
timeout -k 10 300 python bench.py --exchange-rehearsal --no-cpu-baseline > gpurun_out/${TAG}_reh_ar.log 2>gpurun_out/${TAG}_reh_ar.err || { echo "rehearsal failed"; tail -n 20 gpurun_out/${TAG}_reh_ar.err; exit 1; }
timeout -k 10 300 python bench.py --exchange-rehearsal --shard-big --no-cpu-baseline > gpurun_out/${TAG}_reh_rs.log 2>gpurun_out/${TAG}_reh_rs.err || { echo "rehearsal rs failed"; tail -n 20 gpurun_out/${TAG}_reh_rs.err; exit 1; }
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/${TAG}_plain.log 2>/dev/null || { echo "plain failed"; exit 1; }
python - <<PY
import json
for n in ("plain", "reh_ar", "reh_rs"):
    d = json.loads(open("gpurun_out/${TAG}_%s.log" % n).read().strip().splitlines()[-1])
    print(n, d["ms_per_step"], d.get("comm_exposed_ms"), d.get("big_gradients"), d["final_loss"])
PY
