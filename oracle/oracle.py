"""ctypes front-end of the CPU oracle (oracle/dsvt_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of dsvt_oracle.c.  Importable from
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never from the
product package.

Each function takes/returns numpy arrays shaped exactly like the reference
plugin's tensors (SURVEY.md section 8a column 4), batch dim dropped.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

NEG_MASK = np.float32(-3.4028235e38)


def build(force=False):
    so = os.path.join(_HERE, "libdsvt_oracle.so")
    src = os.path.join(_HERE, "dsvt_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libdsvt_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_box_overlap_rows.restype = C.c_float
        _LIB.orc_box_overlap_rows_cr.restype = C.c_float
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class P2FCfg(C.Structure):
    _fields_ = [("max_points_num", C.c_int), ("max_points_num_voxel_filter", C.c_int),
                ("max_pillars_num", C.c_int), ("point_feature_num", C.c_int),
                ("feature_num", C.c_int), ("max_num_points_per_voxel", C.c_int),
                ("min_x", C.c_float), ("max_x", C.c_float), ("min_y", C.c_float),
                ("max_y", C.c_float), ("min_z", C.c_float), ("max_z", C.c_float),
                ("vx", C.c_float), ("vy", C.c_float), ("vz", C.c_float),
                ("gx", C.c_int), ("gy", C.c_int), ("gz", C.c_int)]


class WPCfg(C.Structure):
    _fields_ = [("max_win_num", C.c_int), ("max_voxel_num_per_win", C.c_int),
                ("sparse_x", C.c_int), ("sparse_y", C.c_int), ("sparse_z", C.c_int),
                ("win_x", C.c_int), ("win_y", C.c_int), ("win_z", C.c_int),
                ("shift_x", C.c_int), ("shift_y", C.c_int), ("shift_z", C.c_int),
                ("max_pillars_num", C.c_int)]


class GSCfg(C.Structure):
    _fields_ = [("max_win_num", C.c_int), ("max_voxel_num_per_win", C.c_int),
                ("voxel_num_set", C.c_int), ("win_x", C.c_int), ("win_y", C.c_int),
                ("win_z", C.c_int), ("num_heads", C.c_int)]


def load_data(file_bytes, max_points):
    """include/helper.h:28-72 + src/dsvt-ai-trt.cpp:1909."""
    buf = np.frombuffer(file_bytes, dtype=np.uint8)
    out = np.empty((max_points, 4), np.float32)
    n = C.c_uint32(0)
    rc = lib().orc_load_data(_p(buf), C.c_uint32(buf.size), max_points, _p(out), C.byref(n))
    if rc != 0:
        raise ValueError("num of points exceeds max_points (reference exits)")
    return out, int(n.value)


def points2features(points, n_points, cfg):
    """cfg: dict with the Points2FeaturesPlugin field names."""
    r, v, g = cfg["point_cloud_range"], cfg["voxel_size"], cfg["grid_size"]
    c = P2FCfg(cfg["max_points_num"], cfg["max_points_num_voxel_filter"], cfg["max_pillars_num"],
               cfg["point_feature_num"], cfg["feature_num"], cfg["max_num_points_per_voxel"],
               r[0], r[3], r[1], r[4], r[2], r[5], v[0], v[1], v[2], g[0], g[1], g[2])
    T = cfg["max_num_points_per_voxel"]
    points = np.ascontiguousarray(points, np.float32)
    feat = np.empty((cfg["max_points_num_voxel_filter"], cfg["feature_num"]), np.float32)
    pidx = np.empty((cfg["max_pillars_num"], T), np.uint32)
    coords = np.empty((cfg["max_pillars_num"], 4), np.uint32)
    pcnt = np.empty((cfg["max_pillars_num"], 1), np.uint32)
    P, Nk = C.c_uint32(0), C.c_uint32(0)
    lib().orc_points2features(C.byref(c), _p(points), C.c_uint32(n_points), _p(feat), _p(pidx),
                              _p(coords), _p(pcnt), C.byref(P), C.byref(Nk))
    return dict(feat=feat, pidx=pidx, coords=coords, pcnt=pcnt, P=int(P.value), Nk=int(Nk.value))


def scatter_max(feat, pidx, pcnt, P, max_points_num, max_pillars_num, feature_num):
    feat = np.ascontiguousarray(feat, np.float32)
    T = pidx.shape[-1]
    mp = np.empty((max_points_num, feature_num), np.float32)
    mv = np.empty((max_pillars_num, feature_num), np.float32)
    lib().orc_scatter_max(_p(feat), _p(np.ascontiguousarray(pidx)), _p(np.ascontiguousarray(pcnt)),
                          C.c_uint32(P), max_points_num, max_pillars_num, feature_num, T, _p(mp), _p(mv))
    return mp, mv


def window_partition(coords, P, cfg):
    """cfg: WindowPartitionPlugin fields + max_pillars_num."""
    c = WPCfg(cfg["max_win_num"], cfg["max_voxel_num_per_win"], *cfg["sparse_shape"],
              *cfg["win_shape"], *cfg["shift_list"], cfg["max_pillars_num"])
    MW, Vw, MP = cfg["max_win_num"], cfg["max_voxel_num_per_win"], cfg["max_pillars_num"]
    gidx = np.empty((MW, Vw), np.uint32)
    cinw = np.empty((MW, Vw, 3), np.uint32)
    vcnt = np.empty((MW,), np.uint32)
    c2d = np.empty((MP, 3), np.uint32)
    xy = np.empty((MP, 2), np.float32)
    W = C.c_uint32(0)
    lib().orc_window_partition(C.byref(c), _p(np.ascontiguousarray(coords, np.uint32)), C.c_uint32(P),
                               _p(gidx), _p(cinw), _p(vcnt), C.byref(W), _p(c2d), _p(xy))
    return dict(gidx=gidx, cinw=cinw, vcnt=vcnt, W=int(W.value), c2d=c2d, xy=xy)


def get_set(gidx, cinw, vcnt, W, cfg, num_heads=8):
    """cfg: GetSetPlugin fields."""
    c = GSCfg(cfg["max_win_num"], cfg["max_voxel_num_per_win"], cfg["voxel_num_set"],
              *cfg["win_shape"], num_heads)
    MW, L = cfg["max_win_num"], cfg["voxel_num_set"]
    inds = np.empty((2, MW, L), np.uint32)
    mask = np.empty((2, MW, L), np.float32)
    m0 = np.empty((MW, num_heads, L), np.float32)
    m1 = np.empty((MW, num_heads, L), np.float32)
    S = C.c_uint32(0)
    lib().orc_get_set(C.byref(c), _p(np.ascontiguousarray(gidx)), _p(np.ascontiguousarray(cinw)),
                      _p(np.ascontiguousarray(vcnt)), C.c_uint32(W), _p(inds), _p(mask),
                      C.byref(S), _p(m0), _p(m1))
    return dict(inds=inds, mask=mask, S=int(S.value), mask0_h=m0, mask1_h=m1)


def get_value_by_index(feat, pos, inds, S, axis_id):
    _, MW, L = inds.shape
    Cn = feat.shape[-1]
    q = np.empty((MW, L, Cn), np.float32); k = np.empty_like(q); v = np.empty_like(q)
    lib().orc_get_value_by_index(_p(np.ascontiguousarray(feat, np.float32)),
                                 _p(np.ascontiguousarray(pos, np.float32)),
                                 _p(np.ascontiguousarray(inds)), C.c_uint32(S), MW, L, Cn, axis_id,
                                 _p(q), _p(k), _p(v))
    return q, k, v


def map_set_feature2voxel(set_feat, inds, S, axis_id, max_pillars_num):
    _, MW, L = inds.shape
    Cn = set_feat.shape[-1]
    out = np.empty((max_pillars_num, Cn), np.float32)
    lib().orc_map_set_feature2voxel(_p(np.ascontiguousarray(set_feat, np.float32)),
                                    _p(np.ascontiguousarray(inds)), C.c_uint32(S), MW, L, Cn,
                                    axis_id, max_pillars_num, _p(out))
    return out


def layer_norm(x, P, gamma, beta, eps=0.0):
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(x)
    lib().orc_layer_norm(_p(x), C.c_uint32(P), x.shape[0], x.shape[1], C.c_float(eps),
                         _p(np.ascontiguousarray(gamma, np.float32)),
                         _p(np.ascontiguousarray(beta, np.float32)), _p(out))
    return out


def gelu(x, P):
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(x)
    lib().orc_gelu(_p(x), C.c_uint32(P), x.shape[0], x.shape[1], _p(out))
    return out


def map2bev(feat, coords, P, gx, gy):
    feat = np.ascontiguousarray(feat, np.float32)
    Cn = feat.shape[1]
    bev = np.empty((gy, gx, Cn), np.float32)
    lib().orc_map2bev(_p(feat), _p(np.ascontiguousarray(coords, np.uint32)), C.c_uint32(P), Cn, gx, gy, _p(bev))
    return bev


def filter_box_by_score(scores, classes, xs, ys, center, center_z, angle, dim, cfg):
    """cfg: FilterBoxByScorePlugin fields; point_cloud_range = (xmin,xmax,ymin,ymax,zmin,zmax)."""
    K = cfg["max_top_k"]
    r, v = cfg["point_cloud_range"], cfg["voxel_size"]
    out = np.empty((K, 9), np.float32)
    n = C.c_uint32(0)
    f = lambda a: _p(np.ascontiguousarray(a, np.float32))
    u = lambda a: _p(np.ascontiguousarray(a, np.uint32))
    lib().orc_filter_box_by_score(f(scores), u(classes), u(xs), u(ys), f(center), f(center_z), f(angle),
                                  f(dim), K, C.c_float(r[0]), C.c_float(r[1]), C.c_float(r[2]),
                                  C.c_float(r[3]), C.c_float(r[4]), C.c_float(r[5]),
                                  C.c_float(v[0]), C.c_float(v[1]), C.c_float(cfg["score_threshold"]),
                                  _p(out), C.byref(n))
    return out, int(n.value)


def nms_cpu(boxes9, n, nms_thresh=0.01, trig="ref"):
    """save_result + nms_cpu (include/helper.h:257-283, 470-481).  Returns (rows, keep_idx):
    rows are x,y,z,l,w,h,rt,id,score as save_txt prints them.
    trig = "ref": cosf / sinf / atan2f of the platform libm, the float overloads helper.h resolves to (the reference's arithmetic);
    trig = "cr": each of those values correctly rounded through the double function -- the arithmetic of csrc/nms.hip."""
    boxes9 = np.ascontiguousarray(boxes9[:n], np.float32)
    out = np.empty((max(n, 1), 9), np.float32)
    keep = np.empty((max(n, 1),), np.int32)
    fn = {"ref": lib().orc_nms_cpu, "cr": lib().orc_nms_cpu_cr}[trig]
    k = fn(_p(boxes9), n, C.c_float(nms_thresh), _p(out), _p(keep))
    return out[:k].copy(), keep[:k].copy()


def box_overlap(a9, b9, trig="ref"):
    fn = {"ref": lib().orc_box_overlap_rows, "cr": lib().orc_box_overlap_rows_cr}[trig]
    return float(fn(_p(np.ascontiguousarray(a9, np.float32)), _p(np.ascontiguousarray(b9, np.float32))))


def trig_values(x, y, trig="ref"):
    """cos(x), sin(x), atan2(y, x) as floats: "ref" = cosf / sinf / atan2f of the platform libm, "cr" = correctly rounded through double"""
    x = np.ascontiguousarray(x, np.float32); y = np.ascontiguousarray(y, np.float32)
    c, s_, a = (np.empty_like(x) for _ in range(3))
    lib().orc_trig_values(_p(x), _p(y), len(x), {"ref": 0, "cr": 1}[trig], _p(c), _p(s_), _p(a))
    return c, s_, a


# ---- fingerprints of SURVEY.md section 8(a): FNV-1a 64 over little-endian uint32 stream ----
def fnv1a64(u32):
    data = np.ascontiguousarray(u32, "<u4").tobytes()
    h = 1469598103934665603
    for b in data:
        h ^= b
        h = (h * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return "%016x" % h


def pillar_key_fingerprint(coords, P, gx):
    keys = np.sort(coords[:P, 2].astype(np.uint32) * np.uint32(gx) + coords[:P, 3].astype(np.uint32))
    return fnv1a64(keys)


def set_fingerprint(inds_axis, mask_axis, S, coords, gx):
    """for each set the 36-slot sequence of (cell key, mask<0) pairs; sets sorted lexicographically."""
    key = coords[:, 2].astype(np.uint32) * np.uint32(gx) + coords[:, 3].astype(np.uint32)
    k = key[inds_axis[:S]]                                    # [S, L]
    m = (mask_axis[:S] < 0).astype(np.uint32)
    rows = np.stack([k, m], -1).reshape(S, -1)                # [S, 2L] k0,m0,k1,m1,...
    order = np.lexsort(rows.T[::-1])
    return fnv1a64(rows[order].reshape(-1))
