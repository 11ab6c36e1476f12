#!/bin/bash
# tools/ab_bench.sh NAME...: bench.py with each tools/diag/libsfhip_NAME.so in turn, twice, on this box
for rep in 1 2; do for n in "$@"; do
  SF_LIB=$PWD/tools/diag/libsfhip_$n.so timeout -k 10 150 python bench.py --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$n', d['ms_per_step'], 'wide', r['ms_per_launch'], {k.replace('conv3_dma_','').replace('_kernel',''):v['ms_per_launch'] for k,v in r['other'].items()})"
done; done
