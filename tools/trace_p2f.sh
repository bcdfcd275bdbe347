#!/bin/bash
# per-kernel durations of the voxelizer alone under rocprofv3 (ablate build): tools/trace_p2f.sh <DSVT_P2F_DBG value> [frames]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p2fprof
DSVT_P2F_DBG=$1 DSVT_HIP_LIB=$R/dsvt-ai-trt_amd/libdsvt_hip_ablate.so rocprofv3 --kernel-trace --stats -d /tmp/p2fprof -o t -- python $R/tools/time_p2f.py ${2:-4} > /dev/null 2>&1
python - <<'PY'
import sqlite3, glob
db = glob.glob("/tmp/p2fprof/*results.db")[0]
cur = sqlite3.connect(db).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='view' or type='table'")]
k = [t for t in tabs if t == "kernels"][0]
rows = list(cur.execute(f"select name, count(*), avg(end-start), min(end-start) from {k} where name like '%p2f_%' group by name"))
for n, c, a, m in rows: print(f"  {n[:40]:40s} n={c:3d} avg {a/1e3:7.1f} us  min {m/1e3:7.1f} us")
PY
