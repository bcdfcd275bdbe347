"""Shared test configurations: the reference's compile-time caps (include/params.h) and the
raised caps used for the Waymo-shaped synthetic clouds."""
import os
import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

RANGE_P2F = [-74.88, -74.88, -5.0, 74.88, 74.88, 3.0]     # (xmin,ymin,zmin,xmax,ymax,zmax)
RANGE_FB = [-74.88, 74.88, -74.88, 74.88, -5.0, 3.0]      # (xmin,xmax,ymin,ymax,zmin,zmax)
VOXEL = [0.32, 0.32, 8.0]
GRID = [468, 468, 1]
WINS = [([12, 12, 1], [0, 0, 0]), ([24, 24, 1], [6, 6, 0])]


def caps(name):
    if name == "ref":        # include/params.h:24-27,68-70
        return dict(N=50000, Nk=30000, P=10000, W=800, Vw=576)
    if name == "waymo":      # SURVEY 8a recommended caps (W/S reduced to 2048: S12=1714, S24=1158)
        return dict(N=196608, Nk=196608, P=65536, W=2048, Vw=576)
    if name == "mid":        # config 2: 60k points
        return dict(N=65536, Nk=65536, P=32768, W=2048, Vw=576)
    raise KeyError(name)


def p2f_cfg(c):
    return dict(max_points_num=c["N"], max_points_num_voxel_filter=c["Nk"], max_pillars_num=c["P"],
                point_feature_num=4, feature_num=10, max_num_points_per_voxel=48,
                point_cloud_range=RANGE_P2F, voxel_size=VOXEL, grid_size=GRID)


def wp_cfg(c, i):
    win, shift = WINS[i]
    return dict(max_win_num=c["W"], max_voxel_num_per_win=c["Vw"], sparse_shape=GRID, win_shape=win,
                shift_list=shift, max_pillars_num=c["P"])


def gs_cfg(c, i):
    return dict(max_win_num=c["W"], max_voxel_num_per_win=c["Vw"], voxel_num_set=36, win_shape=WINS[i][0])


def load_frame(name, max_points):
    """reference loadData semantics (include/helper.h:28-72): zero-padded to the cap."""
    raw = np.fromfile(os.path.join(GOLDEN, name + ".bin"), dtype=np.float32).reshape(-1, 4)
    out = np.zeros((max_points, 4), np.float32)
    out[:raw.shape[0]] = raw
    return out, raw.shape[0]


def pad_points(p, max_points):
    out = np.zeros((max_points, 4), np.float32)
    out[:p.shape[0]] = p
    return out, p.shape[0]


def sample_rows(n, seed=0, full_below=24576):
    """Rows on which a per-row fp64 reference is evaluated when the tensor is large (the row-wise kernels -- linears, the encoder MLP -- compute every
    row independently, and the fp64 restatement of 138k rows cost 7 s per case on the GPU box's host cores): None (= all rows) below `full_below`;
    otherwise the first and last 512 rows, ONE row of every 16-row MFMA tile (so a wrong wave tile anywhere is always hit) and 2048 random rows."""
    if n <= full_below:
        return None
    rng = np.random.default_rng(seed + n)
    tiles = np.arange(0, n, 16)
    per_tile = np.minimum(tiles + rng.integers(0, 16, tiles.shape[0]), n - 1)
    return np.unique(np.concatenate([np.arange(512), np.arange(n - 512, n), per_tile, rng.integers(0, n, 2048)]))
