#!/bin/bash
# GPU tests, loss probe, default bench, counter passes, kernel trace.  Every GPU step has its own
# time limit; the chain stops at the first failure.  Database files are deleted before the end
# (the directory that travels back is limited to 64 MiB).
export TMPDIR=/tmp
TAG=${1:-r3b}
mkdir -p gpurun_out
timeout -k 10 1000 python -u -m pytest tests -m gpu -q --timeout 300 --timeout-method thread -p no:cacheprovider > gpurun_out/${TAG}_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/${TAG}_tests.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=" gpurun_out/${TAG}_tests.log | tail -n 25
timeout -k 10 120 python tools/loss_probe.py gpurun_out/${TAG}_loss_probe.json || { echo "loss probe failed"; exit 1; }
timeout -k 10 300 python bench.py > gpurun_out/${TAG}_bench.log 2> gpurun_out/${TAG}_bench.err || { echo "bench failed"; tail -n 20 gpurun_out/${TAG}_bench.err; exit 1; }
cut -c1-900 gpurun_out/${TAG}_bench.log
timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d gpurun_out/${TAG}_pmcA -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_pmcA.log 2>&1 || { echo "pmc pass A failed"; tail -n 5 gpurun_out/${TAG}_pmcA.log; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -d gpurun_out/${TAG}_pmcB -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_pmcB.log 2>&1 || { echo "pmc pass B failed"; tail -n 5 gpurun_out/${TAG}_pmcB.log; exit 1; }
python tools/pmc_mfma_summary.py gpurun_out/${TAG}_pmc_mfma.json gpurun_out/${TAG}_pmcA gpurun_out/${TAG}_pmcB > gpurun_out/${TAG}_pmc_mfma.txt 2>&1
rm -rf gpurun_out/${TAG}_pmcA gpurun_out/${TAG}_pmcB
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/${TAG}_pmcF -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_pmcF.log 2>&1 || { echo "pmc pass F failed"; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/${TAG}_pmcW -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_pmcW.log 2>&1 || { echo "pmc pass W failed"; exit 1; }
python tools/pmc_summary.py gpurun_out/${TAG}_pmcF gpurun_out/${TAG}_pmcW gpurun_out/${TAG}_pmc_hbm_traffic.json > gpurun_out/${TAG}_pmc_hbm.txt 2>&1
rm -rf gpurun_out/${TAG}_pmcF gpurun_out/${TAG}_pmcW
timeout -s KILL 240 rocprofv3 --kernel-trace -d gpurun_out/${TAG}_trace -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_trace.log 2>&1 || { echo "trace failed"; tail -n 5 gpurun_out/${TAG}_trace.log; exit 1; }
python tools/prof_steps.py $(ls -t gpurun_out/${TAG}_trace/*/*.db | head -n 1) 12 gpurun_out/${TAG}_kernel_stats.csv > gpurun_out/${TAG}_steps.txt 2>&1
head -n 12 gpurun_out/${TAG}_steps.txt
rm -rf gpurun_out/${TAG}_trace
du -sh gpurun_out
