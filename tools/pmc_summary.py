"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, csv output) into per-kernel HBM
traffic per launch.  Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md section HBM prescribes:
the counters are in KB (x1024 bytes); on gfx950 FETCH_SIZE counts 128-byte fabric requests at 64 B, i.e. it
reports exactly half of the bytes of a wide (16 B/lane) coalesced read stream -> doubled here; WRITE_SIZE is
uncalibrated and is taken as reported.

    python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_traffic
"""
import collections, csv, json, statistics, sys


def load(d, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f"{d}/pmc_counter_collection.csv")):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg


def main():
    fd, wd, out = sys.argv[1:4]
    f, w = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    res, lines = {}, []
    lines.append(f"# {'launches':>8} {'FETCH_KB(raw)':>14} {'read_MB(x2)':>12} {'WRITE_KB':>12} {'write_MB':>9} {'traffic_MB/launch':>18}  kernel")
    for k in sorted(set(f) | set(w), key=lambda k: -(sum(f.get(k, [0])) + sum(w.get(k, [0])))):
        fm = statistics.mean(f[k]) if k in f else 0.0
        wm = statistics.mean(w[k]) if k in w else 0.0
        rd, wr = 2 * fm * 1024 / 1e6, wm * 1024 / 1e6
        res[k] = dict(launches=len(f.get(k, [])), fetch_kb_raw=round(fm, 1), write_kb=round(wm, 1),
                      read_mb=round(rd, 2), write_mb=round(wr, 2), traffic_mb_per_launch=round(rd + wr, 2))
        lines.append(f"  {len(f.get(k, [])):8d} {fm:14.1f} {rd:12.2f} {wm:12.1f} {wr:9.2f} {rd + wr:18.2f}  {k[:110]}")
    open(out + ".json", "w").write(json.dumps(res, indent=1))
    open(out + ".txt", "w").write("\n".join(lines[:40]) + "\n")
    print("\n".join(lines[:14]))


if __name__ == "__main__":
    main()
