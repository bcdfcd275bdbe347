import sys, numpy as np
a = np.load(sys.argv[1])
for p in sys.argv[2:]:
    b = np.load(p)
    d = a != b
    rows = np.nonzero(d.any(1))[0]
    print(p, "rows differing", rows.tolist(), "of", len(a))
    for r in rows[:6]:
        cols = np.nonzero(d[r])[0]
        print("  row", r, "ncols", len(cols), "first cols", cols[:8].tolist(), "cnt", a[r, -1], b[r, -1], "maxabs", float(np.abs(a[r] - b[r]).max()),
              "a", a[r, cols[:3]].tolist(), "b", b[r, cols[:3]].tolist())
