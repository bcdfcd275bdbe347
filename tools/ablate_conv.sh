#!/bin/bash
# timing ablations of the conv kernels (DSVT_CONV_DBG bits: 1 no halo reload, 2 no weight streaming, 4 no MFMA (8-row kernel only), 8 no store)
for d in 0 1 2 3 8 11; do
  echo "dbg=$d"; DSVT_CONV_DBG=$d python tools/bench_conv.py 2>&1 | grep ours | sed -n '1p;3p;7p' | cut -c1-60
done
