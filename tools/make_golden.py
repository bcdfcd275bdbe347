"""Generates tests/golden/oracle_boxes.npz: the FilterBoxByScore rows of the fp32 CPU oracle (oracle/dense_ref.forward) for every cloud the GPU
parity tests and bench.py check boxes on (tests/golden_oracle.KEYS).  Runs the LIVE oracle (CPU only, no GPU; ~5-40 s per cloud):

    python tools/make_golden.py            # all keys
    python tools/make_golden.py KEY ...    # refresh some keys, keep the others

Each entry stores the rows, the count and what it was made from (md5 of the points, md5 of the weights, the caps); the loader
(tests/golden_oracle.forward) refuses an entry whose inputs differ from the caller's."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402
from tests import golden_oracle as GO  # noqa: E402


def main():
    pkg = G.load_package()
    w = pkg.synth.make_weights()
    keys = sys.argv[1:] or list(GO.KEYS)
    out = {}
    if os.path.exists(GO.BOXES_FILE):
        old = np.load(GO.BOXES_FILE, allow_pickle=False)
        out = {k: old[k] for k in old.files}
    for key in keys:
        caps, pts, n = GO.frame_inputs(pkg, key)
        t0 = time.time()
        rows, cnt = GO.live(pts, n, w, caps)
        assert rows.shape == (500, 9) and not rows[cnt:].any()
        out[key + ".boxes"] = rows[:cnt].copy()
        out[key + ".count"] = np.int32(cnt)
        out[key + ".made_from"] = np.array(GO.fingerprint(pts, n, w, caps))
        print(f"{key}: {cnt} rows, {time.time() - t0:.1f} s", flush=True)
    np.savez_compressed(GO.BOXES_FILE, **out)
    print("wrote", GO.BOXES_FILE, os.path.getsize(GO.BOXES_FILE), "bytes")


if __name__ == "__main__":
    main()
