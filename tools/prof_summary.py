"""Summarise a rocprofv3 --kernel-trace run (rocpd sqlite .db) into the per-kernel table that is
committed under profiles/.  Steady state = the frames between the (first_frame)-th and the last
launch of the voxelizer's first kernel (p2f_partition), so MIOpen's find-mode trials and the warm-up
frames are excluded.

    python tools/prof_summary.py gpurun_out/prof_x/bench_results.db [frames_to_keep] > profiles/xxx.txt
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    keep = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    cur = db.cursor()
    starts = [r[0] for r in cur.execute("select start from kernels where name like '%p2f_partition%' order by start")]
    if len(starts) < keep + 1:
        keep = len(starts) - 1
    t0, t1 = starts[-keep - 1], starts[-1]
    rows = list(cur.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), max(vgpr_count), "
        "max(accum_vgpr_count), max(lds_size), max(grid_x), max(workgroup_x) from kernels where start>=? and start<? "
        "group by name order by 3 desc", (t0, t1)))
    tot = sum(r[2] for r in rows)
    print(f"# rocprofv3 --kernel-trace, steady state: {keep} frames; wall/frame = {(t1 - t0) / keep / 1e6:.3f} ms; "
          f"GPU-busy/frame = {tot / keep / 1e6:.3f} ms")
    print(f"# {'%':>5} {'calls/frame':>11} {'avg_us':>9} {'min_us':>9} {'max_us':>9} {'ms/frame':>9} {'vgpr':>5} {'agpr':>5} {'lds':>7} {'grid':>9} {'wg':>5}  kernel")
    for r in rows:
        print(f"  {r[2] / tot * 100:5.1f} {r[1] / keep:11.1f} {r[3] / 1e3:9.1f} {r[4] / 1e3:9.1f} {r[5] / 1e3:9.1f} "
              f"{r[2] / keep / 1e6:9.3f} {r[6]:5d} {r[7]:5d} {r[8]:7d} {r[9]:9d} {r[10]:5d}  {r[0][:150]}")


if __name__ == "__main__":
    main()
