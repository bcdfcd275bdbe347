"""Committed outputs of the fp32 CPU oracle (oracle/dense_ref.forward: the restatement of the reference network,
src/dsvt-ai-trt.cpp:571-1669 + FilterBoxByScore) on the clouds the GPU parity tests and bench.py's check-cloud leg compare against.

The oracle costs 5-12 s of host time per 180k-point cloud; round 5's GPU suite called it for ~30 clouds and sat 45 s from the driver's limit.
The FilterBoxByScore rows it produces are small (<= 500 x 9 floats), so they are generated ONCE by tools/make_golden.py (which runs the live oracle)
and committed as tests/golden/oracle_boxes.npz.  A fixture entry records what it was made from -- md5 of the cloud's bytes, of the weights and of
the oracle configuration -- and `forward()` REFUSES an entry whose inputs differ from the caller's, so a stale file fails loudly instead of
passing; tests/test_oracle_known_answers.py::test_golden_boxes_are_current re-derives one entry from the live oracle on every CPU run, and
bench.py cross-checks the entry of its frame 0 against the live oracle run of `cpu_baseline.whole_network_port` on the GPU box.

Key scheme (tests/test_pipeline_gpu.py::_frame_and_caps): "000000" / "000003" / "000004" = the reference's .bin frames under the reference caps
(include/params.h:24-27, 68-70); "lidar<N>s<seed>" = synth.lidar_like(N, seed) under the Waymo-sized caps (pipeline.Caps())."""
import hashlib
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BOXES_FILE = os.path.join(GOLDEN, "oracle_boxes.npz")

# every cloud a GPU test or bench.py checks boxes on
KEYS = (["000000", "000003", "000004", "lidar60000s3", "lidar196000s5"] +
        [f"lidar180000s{s}" for s in (0, 1, 3, 7, 9, 16, 21, 23)])

_cache = {}
_wmd5 = {}


def normalise(frame):
    """"lidar180000" -> "lidar180000s0" (the tests' short form of seed 0)"""
    if frame.startswith("lidar") and "s" not in frame[5:]:
        return frame + "s0"
    return frame


def frame_inputs(pkg, key):
    """(caps, zero-padded points [caps.N, 4] float32, n) of a fixture key"""
    from tests import cases
    if key.startswith("lidar"):
        caps = pkg.pipeline.Caps()
        npts, _, seed = key[5:].partition("s")
        pts, n = cases.pad_points(pkg.synth.lidar_like(int(npts), int(seed or 0)), caps.N)
    else:
        caps = pkg.pipeline.Caps.reference()
        pts, n = cases.load_frame(key, caps.N)
    return caps, pts, n


def oracle_cfg(caps):
    from oracle import dense_ref as D
    return D.OracleCfg(max_points=caps.N, max_points_filter=caps.Nk, max_pillars=caps.P, max_win=caps.W, max_vox_per_win=caps.Vw,
                       max_sets=caps.S)


def weights_md5(w):
    k = id(w)
    if k not in _wmd5:
        h = hashlib.md5()
        for name in sorted(w):
            h.update(name.encode()); h.update(np.ascontiguousarray(w[name]).tobytes())
        _wmd5[k] = h.hexdigest()
    return _wmd5[k]


def fingerprint(pts, n, w, caps):
    """what a fixture entry was made from: (md5 of the n live points, md5 of the weights, the oracle caps)"""
    return (hashlib.md5(np.ascontiguousarray(pts[:n]).tobytes()).hexdigest() + ":" + str(int(n)), weights_md5(w),
            f"N={caps.N},Nk={caps.Nk},P={caps.P},W={caps.W},Vw={caps.Vw},S={caps.S}")


def live(pts, n, w, caps):
    """the oracle itself (seconds of host time)"""
    from oracle import dense_ref as D
    eb, ec = D.forward(pts, n, w, oracle_cfg(caps))
    return np.asarray(eb, np.float32), int(ec)


def _load():
    if "npz" not in _cache:
        if not os.path.exists(BOXES_FILE):
            raise FileNotFoundError(f"{BOXES_FILE} is missing: run `python tools/make_golden.py` (runs the live CPU oracle, ~5 min)")
        _cache["npz"] = np.load(BOXES_FILE, allow_pickle=False)
    return _cache["npz"]


def forward(frame, pts, n, w, caps):
    """FilterBoxByScore rows [500, 9] float32 and count of the fp32 oracle for cloud `frame` -- the committed fixture, after checking that it was made
    from exactly these points, weights and caps."""
    key = normalise(frame)
    z = _load()
    if key + ".boxes" not in z.files:
        raise KeyError(f"no golden oracle boxes for {key!r}: add it to tests/golden_oracle.KEYS and run tools/make_golden.py")
    want = fingerprint(pts, n, w, caps)
    have = tuple(str(x) for x in z[key + ".made_from"])
    if have != want:
        raise AssertionError(f"golden oracle boxes of {key!r} are STALE: made from {have}, asked for {want}: re-run tools/make_golden.py")
    cnt = int(z[key + ".count"])
    rows = np.zeros((500, 9), np.float32)
    rows[:cnt] = z[key + ".boxes"]
    return rows, cnt
