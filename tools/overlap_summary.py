"""How busy is the GPU with N frames in flight?  From a rocprofv3 --kernel-trace database: wall time per frame, the UNION of kernel
intervals (time with at least one kernel resident) and the sum of kernel durations, over the steady-state frames.

    python tools/overlap_summary.py bench_results.db [frames_to_keep]
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    keep = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    cur = db.cursor()
    starts = [r[0] for r in cur.execute("select start from kernels where name like '%p2f_partition%' order by start")]
    keep = min(keep, len(starts) - 1)
    t0, t1 = starts[-keep - 1], starts[-1]
    iv = sorted(cur.execute("select start, end from kernels where start>=? and start<?", (t0, t1)))
    tot = sum(e - s for s, e in iv)
    union, cs, ce = 0, None, None
    two = 0                      # time with >= 2 kernels resident
    events = sorted([(s, 1) for s, e in iv] + [(e, -1) for s, e in iv])
    depth, last = 0, None
    for t, d in events:
        if last is not None:
            if depth >= 1: union += t - last
            if depth >= 2: two += t - last
        depth += d; last = t
    wall = t1 - t0
    print(f"frames {keep}: wall/frame {wall / keep / 1e6:.3f} ms, kernel-time sum/frame {tot / keep / 1e6:.3f} ms, "
          f">=1 kernel resident {100.0 * union / wall:.1f} % of wall, >=2 kernels resident {100.0 * two / wall:.1f} % of wall")


if __name__ == "__main__":
    main()
