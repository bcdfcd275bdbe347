"""Keep decisions of the device NMS against the reference's host NMS AT the IoU threshold.

csrc/nms.hip rounds every cos / sin / atan2 value correctly through the double function (oracle: trig="cr"); include/helper.h:117-118,194-195,236-237 gets glibc's
float overloads (trig="ref"), which differ in the last bit of 1.3 % (cosf) / 16 % (sinf) of the values.  On random box sets the two keep lists never differed
(profiles/r05_nms_trig_rates.txt: 0 of 20 000) -- but random pairs sit nowhere near the threshold.  This tool BUILDS pairs at the threshold: for a random pair of rotated
boxes it bisects the centre distance d (a float) to the last value whose reference IoU is still >= 0.01 (nms_cpu suppresses when iou >= nms_thresh, helper.h:257-283), then
walks +-STEPS float neighbours of d -- IoU values within a few ulp of 0.01 on both sides -- and compares the suppress decision of the two arithmetics at every one.

    python tools/nms_threshold_adversaries.py [--pairs 10000] [--steps 8] [--seed 0]"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O

THRESH = np.float32(0.01)
ORC_THRESH = np.float32(1e-8)          # helper.h:257-283's THRESH guard of the union (oracle/dsvt_oracle.c ORC_THRESH)


def decision(a, b, trig):
    """nms_cpu's test for one pair (helper.h:271-277 in float): (suppress?, iou)"""
    so = np.float32(O.box_overlap(a, b, trig))
    sa = np.float32(a[4]) * np.float32(a[3]); sb = np.float32(b[4]) * np.float32(b[3])          # w * l (row: x, y, z, dim0 = l, dim1 = w, ...)
    iou = so / np.maximum(np.float32(np.float32(sa + sb) - so), ORC_THRESH)
    return bool(iou >= THRESH), float(iou)


def pair(rng):
    a = np.zeros(9, np.float32); b = np.zeros(9, np.float32)
    for r in (a, b):
        r[3] = rng.uniform(0.5, 6.0); r[4] = rng.uniform(0.5, 2.5); r[5] = 1.5; r[6] = rng.uniform(-1.57, 1.57); r[8] = 0.5
    a[0], a[1] = rng.uniform(-60, 60), rng.uniform(-60, 60)
    return a, b, rng.uniform(0, 2 * np.pi)


def at_distance(a, b, phi, d):
    b = b.copy(); b[0] = np.float32(a[0] + np.float32(d) * np.float32(np.cos(phi))); b[1] = np.float32(a[1] + np.float32(d) * np.float32(np.sin(phi)))
    return b


def adversaries(pairs, steps, seed):
    rng = np.random.default_rng(seed)
    built = flips_pairs = points = flips_points = 0
    worst = []
    while built < pairs:
        a, b, phi = pair(rng)
        lo, hi = np.float32(0.0), np.float32(12.0)                 # IoU(lo) >= thresh (concentric), IoU(hi) = 0
        if not decision(a, at_distance(a, b, phi, lo), "ref")[0] or decision(a, at_distance(a, b, phi, hi), "ref")[0]:
            continue
        while np.nextafter(lo, np.float32(np.inf)) < hi:            # bisect on the float lattice: lo = a distance that still suppresses, hi = one that does not
            mid = np.float32((np.float64(lo) + np.float64(hi)) / 2)
            if mid <= lo or mid >= hi:
                break
            if decision(a, at_distance(a, b, phi, mid), "ref")[0]: lo = mid
            else: hi = mid
        built += 1
        flipped = False
        d = lo
        for _ in range(steps): d = np.nextafter(d, np.float32(-np.inf))
        for _ in range(2 * steps + 1):
            bb = at_distance(a, b, phi, d)
            r, iou_r = decision(a, bb, "ref"); c, iou_c = decision(a, bb, "cr")
            points += 1
            if r != c:
                flips_points += 1; flipped = True
                worst.append((iou_r, iou_c))
            d = np.nextafter(d, np.float32(np.inf))
        flips_pairs += flipped
    return dict(pairs=built, pairs_with_a_differing_decision=flips_pairs, points=points, differing_points=flips_points,
                rate_per_threshold_point=flips_points / max(points, 1), examples=worst[:5])


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=10000); ap.add_argument("--steps", type=int, default=8); ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    r = adversaries(a.pairs, a.steps, a.seed)
    print(f"{r['pairs']} box pairs bisected onto the IoU threshold 0.01, {2 * a.steps + 1} float neighbours of the centre distance each ({r['points']} threshold points):")
    print(f"  suppress decisions that differ between the reference's trigonometry (glibc float overloads) and the kernel's (correctly rounded): {r['differing_points']} points "
          f"({100 * r['rate_per_threshold_point']:.3f} %) in {r['pairs_with_a_differing_decision']} pairs")
    for e in r["examples"]:
        print(f"  e.g. iou_ref = {e[0]:.10f}, iou_kernel = {e[1]:.10f}")
