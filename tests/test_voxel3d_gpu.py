"""BASELINE configs[4] beyond the partition: ONE DSVT block over the sets of a true 3-D voxel grid.

The reference has no voxel path (its voxel z index is forced to 0: plugins/src/points2Features.cu:689-690,755), but its set
machinery carries z generically (windowPartition.cu:294-301, getSet.cu:386,461) and its encoder layer (src/dsvt-ai-trt.cpp:653-756)
does not care what a voxel is.  So the 3-D half of the config that the oracle CAN pin is: lidar_like(300000, 0) on a 468 x 468 x 32
grid -> 81 090 voxels -> 12 x 12 x 32 windows (up to 1028 voxels) -> 2946 sets of 36 (both sort axes) -> one block = two encoder
layers (QKV with a position-embedding TABLE over the 12 x 12 x 32 window cells, set attention on axis 0 / axis 1, out-proj + LN,
FFN + LN + LN) + the block LayerNorm, in all three precisions, against oracle/dense_ref.dsvt_blocks on the SAME sets, features and
table.
Round 5 adds the rest of SURVEY 8f-4: the stage reduction (csrc/voxel_pool.hip: pooled partition bit-exact against oracle/dense_ref.pool_partition, attention
pooling against dense_ref.stage_reduction_att) and a TWO-STAGE backbone (pipeline3d.Dsvt3dBackbone: 12 x 12 x 32 windows on the 468 x 468 x 32 grid, pooling
by (1, 1, 4), 12 x 12 x 8 windows on the 468 x 468 x 8 grid) against dense_ref.backbone_3d.  None of it exists in the reference: parity unpinned."""
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
    return dict(c=c, P=np_, S_cap=S_cap, rg=rg, x0=x0, tabs=tabs, cell=cell, Pn=vox[4], coords=vox[2], c2d=wpo[4], inds=gso[0], mask=gso[1], Sn=gso[2])


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


# =====================================================================================================================
# Stage reduction + the two-stage backbone (round 5)
# =====================================================================================================================
@pytest.mark.parametrize("stride", [(1, 1, 4), (2, 2, 2), (1, 1, 32)])
def test_voxel_pool_partition_against_oracle(pkg, oracle, grid3d, stride):
    """DsvtVoxelPoolPlugin: pooled coordinates in ascending pooled-cell order, the child table, every voxel's pooled row and the two counts -- bit-exact
    against dense_ref.pool_partition on the 81 090 voxels of the 300k cloud; blob round trip; a capacity that truncates."""
    from oracle import dense_ref as D
    P, g = pkg.plugin, grid3d
    MP = g["c"]["P"]
    coords = g["coords"]
    c2, tab, par = D.pool_partition(host(coords)[0].view(np.int32), g["P"], (468, 468, 32), stride)
    pv = stride[0] * stride[1] * stride[2]
    op = P.add_voxel_pool_op(MP, MP, (468, 468, 32), stride)
    o = op(coords, g["Pn"])
    torch.cuda.synchronize()
    P2 = int(o[3][0])
    assert P2 == len(c2) and int(o[4][0]) == P2 * pv and 0 < P2 <= g["P"]
    assert np.array_equal(host(o[0])[0][:P2].view(np.int32), c2) and not host(o[0])[0][P2:].any()
    assert np.array_equal(host(o[1])[0].view(np.int32)[:P2], tab) and (host(o[1])[0].view(np.int32)[P2:] == -1).all()
    assert np.array_equal(host(o[2])[0, :g["P"], 0].view(np.int32), par) and (host(o[2])[0, g["P"]:, 0].view(np.int32) == -1).all()
    if stride == (1, 1, 4):
        assert P2 < g["P"] and (tab >= 0).sum() == g["P"] and (tab >= 0).sum(1).max() > 1      # several children per pooled voxel do occur
    again = P.Plugin.deserialize("DsvtVoxelPoolPlugin", op.serialize())
    o2 = again(coords, g["Pn"])
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(o, o2))
    small = P.add_voxel_pool_op(MP, 1000, (468, 468, 32), stride)(coords, g["Pn"])           # capacity 1000: the first 1000 pooled voxels, the others' parents -1
    torch.cuda.synchronize()
    assert int(small[3][0]) == 1000 and np.array_equal(host(small[0])[0].view(np.int32), c2[:1000])
    sp = host(small[2])[0, :g["P"], 0].view(np.int32)
    assert np.array_equal(sp, np.where(par < 1000, par, -1))


def _reduction_weights(rng, pv):
    w = {}
    p = "red"
    w[p + ".self_attn.in_proj_weight"] = (rng.standard_normal((3 * C, C)) / np.sqrt(C)).astype(np.float32)
    w[p + ".self_attn.in_proj_bias"] = (rng.standard_normal(3 * C) * 0.02).astype(np.float32)
    w[p + ".self_attn.out_proj.weight"] = (rng.standard_normal((C, C)) / np.sqrt(C)).astype(np.float32)
    w[p + ".self_attn.out_proj.bias"] = (rng.standard_normal(C) * 0.02).astype(np.float32)
    w[p + ".pos_embedding"] = (rng.standard_normal((pv, C)) * 0.5).astype(np.float32)
    w[p + ".norm.weight"] = rng.uniform(0.8, 1.2, C).astype(np.float32); w[p + ".norm.bias"] = (rng.standard_normal(C) * 0.05).astype(np.float32)
    return w


@pytest.mark.parametrize("stride,split", [((1, 1, 4), True), ((1, 1, 4), False), ((2, 2, 2), True)])
def test_stage_reduction_attention_against_oracle(pkg, oracle, grid3d, stride, split):
    """gather (max query, key = x + slot embedding, value = x) -> Q / K / V linears -> one-query attention per pooled voxel over its non-empty slots -> out-proj +
    residual + LayerNorm, as pipeline3d.Dsvt3dBackbone.reduce() wires it, against dense_ref.stage_reduction_att (upstream's Stage_ReductionAtt_Block)."""
    from oracle import dense_ref as D
    P, g = pkg.plugin, grid3d
    MP, np_ = g["c"]["P"], g["P"]
    pv = stride[0] * stride[1] * stride[2]
    rng = np.random.default_rng(7 + pv)
    w = _reduction_weights(rng, pv)
    x = dev(g["x0"][None])
    pool = P.add_voxel_pool_op(MP, MP, (468, 468, 32), stride)(g["coords"], g["Pn"])
    c2, table, par, P2, rows = pool
    ct = P.COMPUTE_SPLIT if split else P.COMPUTE_F32
    scale = np.float32(np.sqrt(C / 8))
    wi, bi = w["red.self_attn.in_proj_weight"].copy(), w["red.self_attn.in_proj_bias"].copy()
    wi[:C] /= scale; bi[:C] /= scale
    src, kin = P.add_pool_gather_op(MP, pv, C, w["red.pos_embedding"])(x, table, P2)
    q = P.add_linear_op(wi[:C], bi[:C], MP, compute_type=ct)(src, P2)[0]
    k = P.add_linear_op(wi[C:2 * C], bi[C:2 * C], MP, compute_type=ct)(kin, g["Pn"])[0]           # K / V per INPUT voxel (the empty slots are masked anyway)
    v = P.add_linear_op(wi[2 * C:], bi[2 * C:], MP, compute_type=ct)(x, g["Pn"])[0]
    ctx = P.add_pool_attention_core_op(MP, pv, C, 8)(q, k, v, table, P2)[0]
    y = P.add_linear_op(w["red.self_attn.out_proj.weight"], w["red.self_attn.out_proj.bias"], MP, layer_norms=[(w["red.norm.weight"], w["red.norm.bias"])], ln_eps=1e-5)(ctx, P2, src)[0]
    torch.cuda.synchronize()
    n2 = int(P2[0])
    ref = D.stage_reduction_att(g["x0"][:np_], host(table)[0].view(np.int32)[:n2], w, "red")
    got = host(y)[0]
    err = np.abs(got[:n2] - ref).max()
    print(f"stage reduction stride {stride} split {split}: {n2} pooled voxels, max |err| {err:.2e}")
    assert err < 1e-4, err
    assert not got[n2:].any()
    # the max query really sees the zero rows of the empty slots: a voxel alone in its pool with negative features gets max(x, 0)
    tab = host(table)[0].view(np.int32)[:n2]
    lone = np.nonzero((tab >= 0).sum(1) == 1)[0][:50]
    s_host = host(src)[0]
    for r in lone:
        assert np.array_equal(s_host[r], np.maximum(g["x0"][tab[r][tab[r] >= 0][0]], 0.0))


def test_two_stage_3d_backbone_against_oracle(pkg, oracle):
    """pipeline3d.Dsvt3dBackbone on lidar_like(300000, 0): 468 x 468 x 32 voxels -> block over 12 x 12 x 32 windows -> pooling by (1, 1, 4) -> block over
    12 x 12 x 8 windows of the 468 x 468 x 8 grid, against dense_ref.backbone_3d: every integer tensor on the way bit-exact (set counts, window
    coordinates, pooled coordinates, child table), features within 3e-4 after each stage (fp32-grade split precision against fp32 on the CPU)."""
    from oracle import dense_ref as D
    w = pkg.synth.make_weights_3d()
    kw = dict(grid=(468, 468, 32), voxel_size=(0.32, 0.32, 0.25), windows=((12, 12, 32), (12, 12, 8)), strides=((1, 1, 4),),
              max_points=327680, max_voxels=98304, max_win=2048, max_sets=4096)
    net = pkg.pipeline3d.Dsvt3dBackbone(w, device=DEV, **kw)
    pts, n = cases.pad_points(pkg.synth.lidar_like(300000, 0), kw["max_points"])
    tr = {}
    x, coords, Pn = net.forward(dev(pts[None]), scalar(n), trace=tr)
    torch.cuda.synchronize()
    otr = {}
    ox, oc = D.backbone_3d(pts, n, w, trace=otr, **kw)
    P0 = int(tr[("in", 0)][2][0]); P1 = int(Pn[0])
    assert P0 == 81090 and P1 == len(oc) and P1 < P0
    for s_ in range(2):
        gx, info = tr[("block", s_)]; rx, rinfo = otr[("block", s_)]
        np_s = len(rx)
        assert int(info["S"][0]) == rinfo["S"] and int(info["W"][0]) == rinfo["W"]
        assert np.array_equal(host(info["inds"])[0], rinfo["inds"]) and np.array_equal(host(info["c2d"])[0][:np_s], rinfo["c2d"][:np_s])
        e = np.abs(host(gx)[0][:np_s] - rx).max()
        print(f"two-stage 3-D backbone, block of stage {s_}: {np_s} voxels, {rinfo['S']} sets, max |err| {e:.2e}")
        assert e < 3e-4, (s_, e)
    px, pc, pn, pinfo = tr[("pool", 0)]; rpx, rc2, rtab, rpar = otr[("pool", 0)]
    assert int(pn[0]) == len(rc2) and np.array_equal(host(pc)[0][:len(rc2)].view(np.int32), rc2)
    assert np.array_equal(host(pinfo["table"])[0].view(np.int32)[:len(rc2)], rtab) and np.array_equal(host(pinfo["parent"])[0, :P0, 0].view(np.int32), rpar)
    assert np.abs(host(px)[0][:len(rc2)] - rpx).max() < 3e-4
    assert np.array_equal(host(coords)[0][:P1].view(np.int32), oc)
    err = np.abs(host(x)[0][:P1] - ox).max()
    print(f"two-stage 3-D backbone: {P0} -> {P1} voxels, final max |err| {err:.2e}")
    assert err < 3e-4 and not host(x)[0][P1:].any()
