#!/bin/bash
# SQ counter pass over one command:  tools/pmc_kernel.sh <tag> <kernel-name-substring> -- <cmd...>
TAG=$1; PAT=$2; shift 3
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE \
  --output-format csv -d $O/p1 -o pmc -- "$@" > $O/p1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES \
  --output-format csv -d $O/p2 -o pmc -- "$@" > $O/p2.log 2>&1
rocprofv3 --pmc TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE --output-format csv -d $O/p3 -o pmc -- "$@" > $O/p3.log 2>&1
cd $R
python - "$O" "$PAT" <<'PY'
import csv, sys, collections, statistics, glob
O, pats = sys.argv[1:3]
for pat in pats.split(","):              # several kernel-name substrings, comma separated
    print("==", pat)
    for p in ("p1", "p2", "p3"):
        agg = collections.defaultdict(list)
        for f in glob.glob(f"{O}/{p}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if pat in r["Kernel_Name"]:
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            print(f"{p} {k:32s} n={len(v):3d} median={statistics.median(v):16.1f}")
        if not agg: print(p, "no rows; log tail:", open(f"{O}/{p}.log").read()[-300:])
PY
