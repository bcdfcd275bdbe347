"""Helpers shared by the end-to-end parity tests and smoke()."""
import numpy as np


def match_boxes(got, n_got, exp, n_exp, tol=1e-3, thr=0.3, thr_band=1e-4):
    """Box rows are only ordered by candidate rank, and two candidates whose scores differ by
    ~1e-6 may swap; so rows are matched by (class, nearest centre) instead of by index.
    Returns (max abs difference over ALL NINE columns of the matched rows -- x, y, z, the three sizes, yaw, class, score --, number of
    unmatched rows that are not within `thr_band` of the score threshold)."""
    got, exp = got[:n_got], exp[:n_exp]
    worst, unmatched = 0.0, 0
    used = np.zeros(n_got, bool)
    for e in exp:
        if n_got == 0:
            unmatched += abs(e[8] - thr) > thr_band
            continue
        d = np.abs(got[:, :2] - e[:2]).max(1) + (got[:, 7] != e[7]) * 1e3 + used * 1e3
        j = int(np.argmin(d))
        if d[j] > tol:
            unmatched += abs(e[8] - thr) > thr_band
            continue
        used[j] = True
        d = np.abs(got[j] - e)
        # yaw = atan(sin / cos) (src/dsvt-ai-trt.cpp:1668-1669) lives in (-pi/2, pi/2): its two ends are one heading.  The wrap applies ONLY there -- both yaws within
        # 0.02 of +-pi/2 --, so a sign / flip error anywhere else still scores as the ~pi it is
        if min(abs(got[j][6]), abs(e[6])) > np.pi / 2 - 0.02:
            d[6] = min(d[6], abs(np.pi - d[6]))
        worst = max(worst, float(d.max()))
    for j in np.nonzero(~used)[0]:
        unmatched += abs(got[j, 8] - thr) > thr_band
    return worst, int(unmatched)
