#!/bin/bash
# A/B builds of csrc/conv.hip: tools/ab_conv.sh <tag> [-DFLAG=V ...]  ->  dsvt-ai-trt_amd/build_ab/libdsvt_<tag>.so (the other objects come from
# build/); run with DSVT_HIP_LIB=dsvt-ai-trt_amd/build_ab/libdsvt_<tag>.so python tools/bench_conv_mx.py ...
set -e
R=$(cd "$(dirname "$0")/.." && pwd)/dsvt-ai-trt_amd
TAG=$1; shift
mkdir -p $R/build_ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-function "$@" -c $R/csrc/conv.hip -o $R/build_ab/conv_$TAG.o
OBJS=$(ls $R/build/*.o | grep -v '/conv.o')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build_ab/libdsvt_$TAG.so $OBJS $R/build_ab/conv_$TAG.o
echo $R/build_ab/libdsvt_$TAG.so
