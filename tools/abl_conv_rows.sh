#!/bin/bash
# timing ablations of conv_rows_kernel (ablation build; wrong results): DSVT_CONV_DBG 1 = no halo requests, 2 = no weight requests, 4 = no epilogue, 8 = no fragment reads
#   bash tools/abl_conv_rows.sh [H cin cout images residual]
export DSVT_HIP_LIB=dsvt-ai-trt_amd/libdsvt_hip_ablate.so
for d in 0 1 2 3 4 7 8 15; do
  echo "DSVT_CONV_DBG=$d"
  DSVT_CONV_DBG=$d python tools/ab_conv_rows.py "$@" 2>&1 | grep "^rows" | tail -2
done
