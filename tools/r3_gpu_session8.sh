#!/bin/bash
# wide weight-gradient kernel: variants of the library on the probe shapes (timing only for the diag builds)
export TMPDIR=/tmp
TAG=${1:-r3i}
mkdir -p gpurun_out
for n in "$@"; do
  [ "$n" = "$TAG" ] && continue
  lib=${n%%:*}; wide=${n##*:}; [ "$wide" = "$n" ] && wide=160
  SF_WGRAD3_WIDE=$wide SF_LIB=$PWD/tools/diag/libsfhip_$lib.so timeout -k 10 120 python tools/wgrad_probe.py 160,200,64,128 160,100,128,256 > gpurun_out/${TAG}_probe_$n.log 2>&1 || { echo "probe $n failed"; tail -n 5 gpurun_out/${TAG}_probe_$n.log; exit 1; }
  echo $n; grep -v amdgpu.ids gpurun_out/${TAG}_probe_$n.log | cut -c1-140
done
