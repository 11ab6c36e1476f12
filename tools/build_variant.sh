#!/bin/bash
# tools/build_variant.sh NAME FILE.hip [-Dflags...]: tools/diag/libsfhip_NAME.so = the library with
# FILE.hip recompiled with the extra flags (A/B timing through the SF_LIB environment variable).
set -e
cd "$(dirname "$0")/.."
CS=tcn-transformer-gan-hybrid-model-for-predicting-streamflow-in-ungauged-basins_amd/csrc
NAME=$1; FILE=$2; shift 2
mkdir -p tools/diag
/opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-value "$@" -c $CS/$FILE -o /tmp/variant_$NAME.o
OBJS=$(ls $CS/*.o | grep -v "${FILE%.hip}.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS /tmp/variant_$NAME.o -o tools/diag/libsfhip_$NAME.so
echo built tools/diag/libsfhip_$NAME.so
