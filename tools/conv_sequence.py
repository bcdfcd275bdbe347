"""Per-launch durations of the dense stage of the LAST forward in a rocprofv3 kernel trace (rocpd .db): python tools/conv_sequence.py x_results.db"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
starts = [r[0] for r in cur.execute("select start from kernels where name like '%p2f_partition%' order by start")]
t0 = starts[-2]; t1 = starts[-1]
rows = list(cur.execute("select name, start, end, grid_x, workgroup_x from kernels where start>=? and start<? order by start", (t0, t1)))
prev = None
for name, s, e, g, w in rows:
    if "conv" in name or "map2bev" in name or "topk" in name:
        short = name.split("(")[0][:70]
        print(f"{(e - s) / 1e3:8.1f} us  idle before {((s - prev) / 1e3 if prev else 0):6.1f}  wgs {g // w:6d}  {short}")
    prev = e
