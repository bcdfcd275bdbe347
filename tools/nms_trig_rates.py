"""How far is the device NMS from the reference's host NMS?  

include/helper.h:117-118,194-195,236-237 call cos / sin / atan2 on floats under libstdc++, i.e. glibc's cosf / sinf / atan2f.  csrc/nms.hip
computes each of those values correctly rounded to float through the double function, and oracle/dsvt_oracle.c restates both arithmetics
(orc_nms_cpu = the reference's overloads, orc_nms_cpu_cr = the kernel's).  This tool measures
  (1) per VALUE: how often glibc's float function differs from the correctly rounded float (this host), and -- with a GPU -- how often the
      device library's own cosf / sinf / atan2f differ from either (why the kernel does not use them);
  (2) per KEEP LIST: over SETS random box sets (clustered so that many boxes overlap, like tests/test_plugins_gpu.py::_nms_boxes) and over the
      bench clouds' own FilterBoxByScore rows if given (--rows file.npy [frames, 500, 9] + counts), how often the two keep lists differ.
python tools/nms_trig_rates.py [--sets 10000] [--boxes 200] [--procs 8] [--gpu]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O


def boxes(rng, n, spread):
    b = np.zeros((n, 9), np.float32)
    b[:, 0:2] = rng.uniform(-spread, spread, (n, 2)); b[:, 2] = rng.uniform(-2, 2, n)
    b[:, 3] = rng.uniform(0.5, 6.0, n); b[:, 4] = rng.uniform(0.5, 2.5, n); b[:, 5] = rng.uniform(1, 3, n)
    b[:, 6] = rng.uniform(-1.57, 1.57, n); b[:, 7] = rng.integers(0, 10, n)
    b[:, 8] = np.sort(rng.uniform(0.3, 1.0, n).astype(np.float32))[::-1]
    return b


def work(args):
    lo, hi, nb = args
    diff = pairs = 0
    worst = []
    for s in range(lo, hi):
        rng = np.random.default_rng(1000 + s)
        b = boxes(rng, nb, (8.0, 15.0, 40.0)[s % 3])
        _, k0 = O.nms_cpu(b, nb, 0.01, trig="ref")
        _, k1 = O.nms_cpu(b, nb, 0.01, trig="cr")
        if not np.array_equal(k0, k1):
            diff += 1; worst.append((s, len(k0), len(k1)))
    return diff, worst


def value_rates(n=4_000_000, seed=0, gpu=False):
    rng = np.random.default_rng(seed)
    x = rng.uniform(-np.pi, np.pi, n).astype(np.float32)
    y = (rng.standard_normal(n) * 3).astype(np.float32); xa = (rng.standard_normal(n) * 3).astype(np.float32)
    cr, sr, _ = O.trig_values(x, y, "ref"); cc, sc, _ = O.trig_values(x, y, "cr")
    _, _, ar = O.trig_values(xa, y, "ref"); _, _, ac = O.trig_values(xa, y, "cr")
    out = dict(n=n, glibc_cosf_ne_cr=float((cr != cc).mean()), glibc_sinf_ne_cr=float((sr != sc).mean()), glibc_atan2f_ne_cr=float((ar != ac).mean()))
    if gpu:
        import torch
        d = torch.device("cuda:0")
        tx, ty, txa = (torch.from_numpy(v).to(d) for v in (x, y, xa))
        dc, ds, da = torch.cos(tx).cpu().numpy(), torch.sin(tx).cpu().numpy(), torch.atan2(ty, txa).cpu().numpy()
        dcc = torch.cos(tx.double()).float().cpu().numpy(); dsc = torch.sin(tx.double()).float().cpu().numpy()
        dac = torch.atan2(ty.double(), txa.double()).float().cpu().numpy()
        out.update(device_f32_cos_ne_glibc=float((dc != cr).mean()), device_f32_sin_ne_glibc=float((ds != sr).mean()), device_f32_atan2_ne_glibc=float((da != ar).mean()),
                   device_cr_cos_ne_glibc=float((dcc != cr).mean()), device_cr_sin_ne_glibc=float((dsc != sr).mean()), device_cr_atan2_ne_glibc=float((dac != ar).mean()),
                   device_cr_cos_ne_host_cr=float((dcc != cc).mean()), device_cr_sin_ne_host_cr=float((dsc != sc).mean()), device_cr_atan2_ne_host_cr=float((dac != ac).mean()))
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--sets", type=int, default=10000); ap.add_argument("--boxes", type=int, default=200)
    ap.add_argument("--procs", type=int, default=max(1, (os.cpu_count() or 2) - 1)); ap.add_argument("--gpu", action="store_true")
    ap.add_argument("--rows", default=None, help="npz with rows [F, 500, 9] and counts [F]: FilterBoxByScore outputs of real frames")
    a = ap.parse_args()
    print("per value:", {k: (v if k == "n" else float(f"{v:.3g}")) for k, v in value_rates(gpu=a.gpu).items()}, flush=True)
    # per IoU: overlapping pairs of the first sets, overlap area both ways
    nd = npair = 0; worst_rel = 0.0
    for s_ in range(40):
        b = boxes(np.random.default_rng(1000 + s_), 120, 8.0)
        for i in range(120):
            for j in range(i + 1, 120):
                o0, o1 = O.box_overlap(b[i], b[j], "ref"), O.box_overlap(b[i], b[j], "cr")
                if o0 > 0 or o1 > 0:
                    npair += 1; nd += o0 != o1; worst_rel = max(worst_rel, abs(o0 - o1) / max(o0, o1))
    print(f"per overlap: {nd} of {npair} overlapping pairs differ in the area's bits ({nd / max(npair, 1):.3f}), worst relative difference {worst_rel:.2e}", flush=True)
    t0 = time.time()
    import multiprocessing as mp
    chunks = [(i * a.sets // a.procs, (i + 1) * a.sets // a.procs, a.boxes) for i in range(a.procs)]
    with mp.Pool(a.procs) as pool:
        res = pool.map(work, chunks)
    diff = sum(r[0] for r in res); cases = [c for r in res for c in r[1]]
    print(f"keep lists: {diff} of {a.sets} random sets of {a.boxes} boxes differ between the reference's float overloads and the correctly rounded "
          f"values ({diff / a.sets:.2e}); {time.time() - t0:.0f} s on {a.procs} processes", flush=True)
    if cases: print("  differing sets (seed, kept ref, kept cr):", cases[:20])
    if a.rows:
        z = np.load(a.rows)
        d = 0
        for r, c in zip(z["rows"], z["counts"]):
            d += not np.array_equal(O.nms_cpu(r, int(c), 0.01, trig="ref")[1], O.nms_cpu(r, int(c), 0.01, trig="cr")[1])
        print(f"frames: {d} of {len(z['counts'])} FilterBoxByScore outputs give different keep lists")
