#!/bin/bash
# bash tools/ab_variants.sh "<tags>" <ab_conv_rows.py args>: the rows kernel of every variant library (dsvt-ai-trt_amd/variants/), two rounds, same box
TAGS=$1; shift
for rep in 1 2; do
  echo "product: $(python tools/ab_conv_rows.py "$@" 2>&1 | grep '^rows' | tail -1)"
  for t in $TAGS; do
    echo "$t: $(DSVT_HIP_LIB=dsvt-ai-trt_amd/variants/libdsvt_hip_$t.so python tools/ab_conv_rows.py "$@" 2>&1 | grep '^rows' | tail -1)"
  done
done
