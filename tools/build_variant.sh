#!/bin/bash
# A/B builds of ONE translation unit with extra -D flags, linked against the product objects:
#   bash tools/build_variant.sh <tag> <source.hip> [-DFLAG=..]...   ->  dsvt-ai-trt_amd/variants/libdsvt_hip_<tag>.so   (load it with DSVT_HIP_LIB)
set -e
TAG=$1; SRC=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd); P=$R/dsvt-ai-trt_amd
mkdir -p $P/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function "$@" -c $P/csrc/$SRC -o $P/variants/${SRC%.hip}_$TAG.o
OBJS=$(ls $P/build/*.o | grep -v "/${SRC%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/variants/libdsvt_hip_$TAG.so $OBJS $P/variants/${SRC%.hip}_$TAG.o
echo $P/variants/libdsvt_hip_$TAG.so
