#!/bin/bash
# timing ablations of the halo conv kernel (DSVT_CONV_DBG bits: 1 no halo reload, 2 no weight streaming, 4 no MFMA, 8 no store)
for d in 0 1 2 3 4 8 7 11 15; do
  echo -n "dbg=$d  "; DSVT_CONV_TH=${TH:-8} DSVT_CONV_DBG=$d python tools/bench_conv.py 2>&1 | grep ours | head -1 | cut -c1-60
done
