"""Average / minimum duration per kernel name of a rocprofv3 --kernel-trace database (any script, not only bench.py: prof_summary.py wants the
voxelizer's launches as frame marks).  python tools/kernel_avgs.py <results.db> [substring ...]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
pats = sys.argv[2:]
for name, cnt, avg, mn in db.execute("select name, count(*), avg(end-start)/1e3, min(end-start)/1e3 from kernels group by name order by 3 desc"):
    if not pats or any(p in name for p in pats):
        print(f"{avg:9.1f} us avg {mn:9.1f} us min  x{cnt:5d}  {name[:110]}")
