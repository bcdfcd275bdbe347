#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the kernels whose name contains <pattern> over one command (two separate --pmc passes):
#   tools/pmc_fetch_kernel.sh <tag> <pattern> -- <cmd...>        (FETCH_SIZE in KB as the counter reports it; the guide's gfx950 correction doubles the read side)
TAG=$1; PAT=$2; shift 3
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmcf_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -o pmc -- "$@" > $O/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -o pmc -- "$@" > $O/w.log 2>&1
cd $R
python - "$O" "$PAT" <<'PY'
import csv, sys, collections, statistics, glob
O, pat = sys.argv[1:3]
for p in ("f", "w"):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"{O}/{p}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                agg[(r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(f"{k[0]:40s} {k[1]:12s} n={len(v):3d} median={statistics.median(v):12.1f} KB")
PY
