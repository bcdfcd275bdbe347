# ablation runs of the conv microbench on the -DDSVT_ABLATE A/B build (tools/ab_conv.sh v8 ... -DDSVT_ABLATE)
for n in 256 192 128 64; do for d in 0 8; do echo "== ncu $n dbg $d"; DSVT_CONV_NCU=$n DSVT_CONV_DBG=$d DSVT_HIP_LIB=dsvt-ai-trt_amd/build_ab/libdsvt_v8.so python tools/bench_conv_mx.py 4 mx,f16 2>&1 | grep "468x468 128->128 res=0 x1\|dense-stage"; done; done
