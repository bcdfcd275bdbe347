"""BASELINE configs[4] beyond the partition: ONE DSVT block over the sets of a true 3-D voxel grid.

The reference has no voxel path (its voxel z index is forced to 0: plugins/src/points2Features.cu:689-690,755), but its set
machinery carries z generically (windowPartition.cu:294-301, getSet.cu:386,461) and its encoder layer (src/dsvt-ai-trt.cpp:653-756)
does not care what a voxel is.  So the 3-D half of the config that the oracle CAN pin is: lidar_like(300000, 0) on a 468 x 468 x 32
grid -> 81 090 voxels -> 12 x 12 x 32 windows (up to 1028 voxels) -> 2946 sets of 36 (both sort axes) -> one block = two encoder
layers (QKV with a position-embedding TABLE over the 12 x 12 x 32 window cells, set attention on axis 0 / axis 1, out-proj + LN,
FFN + LN + LN) + the block LayerNorm, in all three precisions, against oracle/dense_ref.dsvt_blocks on the SAME sets, features and
table.  (3-D pooling / multi-stage set_info have no reference and are not built.)"""
import numpy as np
import pytest
import torch

from tests import cases
from tests.test_plugins_gpu import dev, host, scalar

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
C = 192


@pytest.fixture(scope="module")
def grid3d(pkg, oracle):
    P, O = pkg.plugin, oracle
    c = dict(N=327680, Nk=327680, P=98304, W=2048, Vw=4608)
    grid, vox_size, win = [468, 468, 32], [0.32, 0.32, 0.25], [12, 12, 32]
    pts, n = cases.pad_points(pkg.synth.lidar_like(300000, 0), c["N"])
    ref = O.points2features(pts, n, dict(cases.p2f_cfg(c), voxel_size=vox_size, grid_size=grid))
    S_cap = 4096
    wcfg = dict(max_win_num=c["W"], max_voxel_num_per_win=c["Vw"], sparse_shape=grid, win_shape=win, shift_list=[0, 0, 0], max_pillars_num=c["P"])
    rw = O.window_partition(ref["coords"], ref["P"], wcfg)
    rg = O.get_set(rw["gidx"], rw["cinw"], rw["vcnt"], rw["W"], dict(max_win_num=S_cap, max_voxel_num_per_win=c["Vw"], voxel_num_set=36, win_shape=win))
    assert ref["P"] == 81090 and rg["S"] == 2946 and rw["vcnt"].max() == 1028
    # the same tensors from the plugins (bit-exact against the oracle: tests/test_plugins_gpu.py::test_config4_300k_cloud_3d_voxel_grid)
    vox = P.add_voxel_generator(c["N"], c["Nk"], c["P"], 4, 10, 48, -74.88, 74.88, -74.88, 74.88, -5.0, 3.0, *vox_size, *grid)(dev(pts[None]), scalar(n))
    wpo = P.add_window_partition(c["W"], c["Vw"], *grid, *win, 0, 0, 0)(vox[2], vox[4])
    gso = P.add_get_set_op(c["W"], c["Vw"], 36, *win, max_set_num=S_cap)(wpo[0], wpo[1], wpo[2], wpo[3])
    torch.cuda.synchronize()
    assert np.array_equal(host(gso[0])[0], rg["inds"]) and int(gso[2][0]) == rg["S"] and np.array_equal(host(wpo[4])[0], rw["c2d"])
    rng = np.random.default_rng(42)
    np_ = ref["P"]
    x0 = np.zeros((c["P"], C), np.float32); x0[:np_] = rng.standard_normal((np_, C))
    tabs = [(rng.standard_normal((12 * 12 * 32, C)) * 0.5).astype(np.float32) for _ in range(2)]       # one table per encoder layer
    cell = (rw["c2d"][:np_, 0].astype(np.int64) * 12 + rw["c2d"][:np_, 1]) * 12 + rw["c2d"][:np_, 2]   # (z * wy + y) * wx + x
    assert cell.max() < 12 * 12 * 32 and len(np.unique(rw["c2d"][:np_, 0])) > 8                        # z really varies inside the windows
    return dict(c=c, P=np_, S_cap=S_cap, rg=rg, x0=x0, tabs=tabs, cell=cell, Pn=vox[4], c2d=wpo[4], inds=gso[0], mask=gso[1], Sn=gso[2])


def _oracle_block(pkg, g, w):
    from oracle import dense_ref as D
    MP = g["c"]["P"]
    pe = {}
    for l in range(2):
        full = np.zeros((MP, C), np.float32); full[:g["P"]] = g["tabs"][l][g["cell"]]
        pe[(0, l)] = full
    st = dict(vfeat=g["x0"], P=g["P"], gss=[g["rg"], g["rg"]], pe=pe)
    cfg = D.OracleCfg(max_pillars=MP, blocks=1)
    tr = {}
    out = D.dsvt_blocks(st, w, cfg, nblocks=1, trace=tr)
    return out, tr


def _layer_weights(w, l):
    lp = f"module.backbone_3d.stage_0.0.encoder_list.{l}"
    wi = w[lp + ".win_attn.self_attn.in_proj_weight"].copy(); bi = w[lp + ".win_attn.self_attn.in_proj_bias"].copy()
    scale = np.float32(np.sqrt(C / 8))
    wi[:C] /= scale; bi[:C] /= scale                     # Q / sqrt(head_dim) after the bias (src/dsvt-ai-trt.cpp:386-405)
    ln = lambda n: (w[lp + n + ".weight"], w[lp + n + ".bias"])
    lns = [ln(".win_attn.norm1"), ln(".win_attn.norm2"), ln(".norm")]
    if l == 1:
        lns.append((w["module.backbone_3d.residual_norm_stage_0.0.weight"], w["module.backbone_3d.residual_norm_stage_0.0.bias"]))
    return lp, wi, bi, lns


@pytest.mark.parametrize("mode", ["f32", "split", "f16"])
def test_one_dsvt_block_over_3d_voxel_sets(pkg, oracle, grid3d, mode):
    P = pkg.plugin
    g = grid3d
    w = pkg.synth.make_weights(with_bev=False)
    ref, tr = _oracle_block(pkg, g, w)
    MP, np_ = g["c"]["P"], g["P"]
    x = dev(g["x0"][None]); xb = x
    xh = x.half() if mode == "f16" else None
    for l in range(2):
        lp, wi, bi, lns = _layer_weights(w, l)
        mk = lambda k: w[lp + k]
        mlp_w = (mk(".win_attn.self_attn.out_proj.weight"), mk(".win_attn.self_attn.out_proj.bias"), mk(".win_attn.linear1.weight"),
                 mk(".win_attn.linear1.bias"), mk(".win_attn.linear2.weight"), mk(".win_attn.linear2.bias"))
        tab = dev(g["tabs"][l][None])
        if mode == "f32":
            # the unfused reference wiring on exact fp32 MFMA: materialised position rows, three linears with fused epilogues
            pos = torch.zeros((1, MP, C), device=DEV); pos[0, :np_] = tab[0][torch.from_numpy(g["cell"]).to(DEV)]
            qkv = P.add_linear_op(wi, bi, MP, add_cols=2 * C)(x, g["Pn"], pos)[0]
            att = P.add_set_attention_op(g["S_cap"], 36, C, 8, l, MP)(qkv, g["inds"], g["mask"], g["Sn"])[0]
            s1 = P.add_linear_op(mlp_w[0], mlp_w[1], MP, layer_norms=lns[:1])(att, g["Pn"], x)[0]
            h = P.add_linear_op(mlp_w[2], mlp_w[3], MP, activation=P.ACT_GELU)(s1, g["Pn"])[0]
            fc2 = P.add_linear_op(mlp_w[4], mlp_w[5], MP, layer_norms=lns[1:])
            x = (fc2(h, g["Pn"], s1, x, xb) if l == 1 else fc2(h, g["Pn"], s1, x))[0]
        elif mode == "split":
            qkv = P.add_linear_op(wi, bi, MP, add_cols=2 * C, compute_type=P.COMPUTE_SPLIT, add_gather_width=12, add_gather_height=12)(x, g["Pn"], tab, g["c2d"])[0]
            att = P.add_set_attention_op(g["S_cap"], 36, C, 8, l, MP)(qkv, g["inds"], g["mask"], g["Sn"])[0]
            mlp = P.add_encoder_mlp_op(*mlp_w, lns, MP, split_precision=True)
            x = (mlp(att, g["Pn"], x, xb) if l == 1 else mlp(att, g["Pn"], x))[0]
        else:
            qkv = P.add_linear_op(wi, bi, MP, add_cols=2 * C, compute_type=P.COMPUTE_F16, input_half=True, output_mode=P.OUT_F16,
                                  add_gather_width=12, add_gather_height=12)(xh, g["Pn"], tab.half(), g["c2d"])[0]
            att = P.add_set_attention_op(g["S_cap"], 36, C, 8, l, MP, io_half=True)(qkv, g["inds"], g["mask"], g["Sn"])[0]
            mlp = P.add_encoder_mlp_op(*mlp_w, lns, MP)
            x, xh = mlp(att, g["Pn"], x, xb) if l == 1 else mlp(att, g["Pn"], x)
        torch.cuda.synchronize()
        if l == 0:
            e0 = np.abs(host(x)[0][:np_] - tr[(0, 0)][:np_]).max()
            assert e0 < (1e-4 if mode != "f16" else 2e-2), (mode, e0)
    got = host(x)[0]
    err = np.abs(got[:np_] - ref[:np_])
    print(f"3-D block, mode {mode}: max |err| {err.max():.3e}, mean {err.mean():.3e} over {np_} voxels x 192 channels")
    if mode == "f16":
        assert err.max() < 3e-2 and err.mean() < 1.5e-3, (err.max(), err.mean())      # fp16 operands: 2^-11 per operand through two layers of O(1) LayerNorm outputs
    else:
        assert err.max() < 2e-4, err.max()                                             # the fp32 bar of the pillar model's per-block check
    assert not got[np_:].any()
