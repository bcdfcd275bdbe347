ulimit -c 0
cd $GRAFT_REPO_ROOT
run() { # $1 = tag, rest = env
  for k in 1 2 3 4 5 6; do
    (env "${@:2}" SECONDS=10 python tools/dbg_two_proc.py 0 > /tmp/o0.txt 2>&1 &) ; env "${@:2}" SECONDS=10 python tools/dbg_two_proc.py 1 > /tmp/o1.txt 2>&1; sleep 2
    echo "$1 run $k: $(grep '^rank' /tmp/o0.txt | tail -1 | cut -c1-160) || $(grep '^rank' /tmp/o1.txt | tail -1 | cut -c1-160)"
  done
}
run rows X=1
run wide DSVT_HIP_LIB=dsvt-ai-trt_amd/libdsvt_hip_ablate.so DSVT_CONV_ROWS=0
