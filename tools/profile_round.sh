#!/bin/bash
# Collect the rocprofv3 evidence for one bench configuration on the GPU box (run through gpurun):
#   tools/profile_round.sh <tag> [extra bench.py flags]
# writes gpurun_out/prof_<tag>/{trace,pmc_fetch,pmc_write}; kernel trace and the two PMC passes are separate runs.
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 24 --warmup 3 --repeats 1 --no-cpu-baseline --no-latency-mode --no-parity-mode --no-other-configs --no-cpp-host $*"
rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- $B > $O/trace.log 2>&1
grep '^{' $O/trace.log | tail -1 > $O/bench_line.json
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o pmc -- $B --no-graph --streams 1 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o pmc -- $B --no-graph --streams 1 > $O/pmc_write.log 2>&1
cd $R
python tools/prof_summary.py $O/trace/bench_results.db > $O/kernel_trace.txt 2>&1
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic > /dev/null 2>&1
rm -rf $O/pmc_fetch/*.db $O/trace/*.db 2>/dev/null
head -30 $O/kernel_trace.txt | cut -c1-150
