#!/bin/bash
# GPU tests, default bench, counter passes, kernel trace.
# Every GPU step has its own time limit; the chain stops at the first failure.
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 1000 python -u -m pytest tests -m gpu -q --timeout 300 --timeout-method thread -p no:cacheprovider > gpurun_out/r3_tests1.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3_tests1.log
tail -n 25 gpurun_out/r3_tests1.log
timeout -k 10 300 python bench.py > gpurun_out/r3_bench1.log 2> gpurun_out/r3_bench1.err || { echo "bench failed"; tail -n 20 gpurun_out/r3_bench1.err; exit 1; }
cat gpurun_out/r3_bench1.log
timeout -s KILL 60 rocprofv3 -L > gpurun_out/r3_counters.txt 2>&1 || echo "counter list failed"
grep -c . gpurun_out/r3_counters.txt
grep -E "SQ_VALU_MFMA_BUSY_CYCLES|SQ_WAIT_INST_ANY|SQ_LDS_IDX_ACTIVE|GRBM_GUI_ACTIVE|SQ_BUSY_CYCLES" gpurun_out/r3_counters.txt | cut -c1-160 | sort -u | tail -n 12
timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d gpurun_out/r3_pmcA -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r3_pmcA.log 2>&1 || { echo "pmc pass A failed"; tail -n 5 gpurun_out/r3_pmcA.log; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -d gpurun_out/r3_pmcB -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r3_pmcB.log 2>&1 || { echo "pmc pass B failed"; tail -n 5 gpurun_out/r3_pmcB.log; exit 1; }
python tools/pmc_mfma_summary.py gpurun_out/r3_pmc_mfma.json gpurun_out/r3_pmcA gpurun_out/r3_pmcB > gpurun_out/r3_pmc_mfma.txt 2>&1
head -n 24 gpurun_out/r3_pmc_mfma.txt
timeout -s KILL 240 rocprofv3 --kernel-trace -d gpurun_out/r3_trace -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/r3_trace.log 2>&1 || { echo "trace failed"; tail -n 5 gpurun_out/r3_trace.log; exit 1; }
python tools/prof_steps.py $(ls -t gpurun_out/r3_trace/*/*.db | head -n 1) 12 gpurun_out/r3a_kernel_stats.csv > gpurun_out/r3a_steps.txt 2>&1
head -n 40 gpurun_out/r3a_steps.txt
rm -rf gpurun_out/r3_trace gpurun_out/r3_pmcA gpurun_out/r3_pmcB
