"""Per-kernel average duration of the last frames of a rocprofv3 --kernel-trace run:  python tools/kernel_times.py <db> [pattern]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); pat = sys.argv[2] if len(sys.argv) > 2 else ""
cur = db.cursor()
starts = [r[0] for r in cur.execute("select start from kernels where name like '%p2f_partition%' order by start")]
t0, t1 = starts[-4], starts[-1]
rows = list(cur.execute("select name, count(*), avg(end-start), sum(end-start) from kernels where start>=? and start<? group by name order by 4 desc", (t0, t1)))
tot = sum(r[3] for r in rows)
print(f"GPU-busy/frame {tot / 3 / 1e6:.3f} ms; wall/frame {(t1 - t0) / 3 / 1e6:.3f} ms")
for r in rows:
    if pat in r[0]:
        print(f"{r[1] / 3:6.1f} x {r[2] / 1e3:8.1f} us = {r[3] / 3 / 1e6:7.3f} ms  {r[0][:90]}")
