import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
pkg = G.load_package(); P = pkg.plugin
from oracle import oracle as O
from tests import cases
from tests.test_plugins_gpu import dev, scalar
c = cases.caps("ref")
pts, n = cases.load_frame("000000", c["N"])
vox = O.points2features(pts, n, cases.p2f_cfg(c))
rw = O.window_partition(vox["coords"], vox["P"], cases.wp_cfg(c, 0))
gs = O.get_set(rw["gidx"], rw["cinw"], rw["vcnt"], rw["W"], cases.gs_cfg(c, 0))
rng = np.random.default_rng(100)
Pn = vox["P"]
inds = gs["inds"][0]
for trial, (qs, vmode) in enumerate([(0.0, "rand"), (0.0, "ones"), (0.12, "ones"), (0.0, "key")]):
    qkv = np.zeros((c["P"], 576), np.float32)
    base = rng.standard_normal((Pn, 576)).astype(np.float32)
    qkv[:Pn, :192] = base[:, :192] * qs; qkv[:Pn, 192:384] = base[:, 192:384] * 1.5
    if vmode == "rand": qkv[:Pn, 384:] = base[:, 384:]
    elif vmode == "ones": qkv[:Pn, 384:] = 1.2345678
    else: qkv[:Pn, 384:] = (np.arange(Pn, dtype=np.float32)[:, None] * 0.001 + 0.5)
    args = (dev(qkv[None]), dev(gs["inds"][None]), dev(gs["mask"][None]), scalar(gs["S"]))
    got = P.add_set_attention_op(c["W"], 36, 192, 8, 0, c["P"], split_precision=True)(*args)[0][0].cpu().numpy()
    exact = P.add_set_attention_op(c["W"], 36, 192, 8, 0, c["P"])(*args)[0][0].cpu().numpy()
    e = np.abs(got[:Pn] - exact[:Pn])
    bad = sorted(set((int(a), int(b) // 24) for a, b in np.argwhere(e > 2e-5)))
    print(f"trial {trial} q-scale {qs} V {vmode}: max |split - exact| {e.max():.2e} mean {e.mean():.2e} bad pairs {bad[:12]}")
    if bad:
        rr, hh = bad[0]
        print("   got", got[rr, hh * 24:hh * 24 + 6], "exact", exact[rr, hh * 24:hh * 24 + 6])
