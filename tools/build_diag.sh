#!/bin/bash
# Builds tools/diag/libsfhip_stamps.so: the library with sf_conv3x3.hip compiled with in-kernel
# shader-clock stamps (-DSF_CONV3_STAMPS), for tools/conv_stamps.py.  Extra -D flags may be passed.
set -e
cd "$(dirname "$0")/.."
CS=tcn-transformer-gan-hybrid-model-for-predicting-streamflow-in-ungauged-basins_amd/csrc
make -C $CS -j8 > /dev/null
mkdir -p tools/diag
/opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-value -DSF_CONV3_STAMPS "$@" -c $CS/sf_conv3x3.hip -o /tmp/sf_conv3x3_stamps.o
OBJS=$(ls $CS/*.o | grep -v sf_conv3x3.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS /tmp/sf_conv3x3_stamps.o -o tools/diag/libsfhip_stamps.so
echo built tools/diag/libsfhip_stamps.so
