"""Kernel-level parity of the fp16-mode kernels bench.py actually runs -- set_attention_f16_kernel, encoder_mlp_stream_kernel,
pfn_kernel, linear_f16_rows_kernel -- each against the ORACLE's arithmetic (oracle/dense_ref.py restatement of the reference's
TensorRT layers, oracle/dsvt_oracle.c restatement of its plugins) evaluated on the SAME fp16-rounded operands.

The reference has no fp16 vectors (its kFP16 build is TensorRT's choice per layer, include/params.h:332), so the statement
tested here is: "fp16 operands, fp32 accumulate, fp32 LayerNorm / softmax" reproduces the reference's fp32 arithmetic on
those operands up to (a) fp32 summation order and (b) the fp16 roundings of intermediates the kernel itself performs, which
each reference below restates explicitly.  A dropped k-chunk, a mis-masked key or a stale LDS column moves the result by
O(0.1); the bounds are 1e-3-class.
"""
import numpy as np
import pytest
import torch

from tests import cases
from tests.test_plugins_gpu import dev, host, scalar, make_voxelizer

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def r16(a):
    """round to fp16 and back (numpy, round-to-nearest-even like v_cvt_f16_f32)"""
    return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)


# =====================================================================================================================
# set_attention_f16_kernel (DsvtSetAttentionPlugin io_half=1) vs GetValueByIndex -> multHeadAttention core -> MapSetFeature2Voxel
# =====================================================================================================================
def _attention_reference(O, qkv16, gs, axis, max_pillars):
    """oracle gather (getValueByIndex.cu:282-355) -> dense_ref.mha with identity projections (= the attention core of
    src/dsvt-ai-trt.cpp:352-417 on already projected q, k, v) -> oracle scatter (mapSetFeature2voxel.cu:258-320)"""
    from oracle import dense_ref as D
    C = 192
    S = gs["S"]
    zero = np.zeros_like(qkv16[:, :C])
    sq24 = np.float32(np.sqrt(24.0))
    # get_value_by_index returns (feat + pos, feat + pos, feat)[inds]: call it once per operand with pos = 0
    q = O.get_value_by_index(qkv16[:, :C] * sq24, zero, gs["inds"], S, axis)[2]          # D.mha divides Q by sqrt(24)
    k = O.get_value_by_index(qkv16[:, C:2 * C], zero, gs["inds"], S, axis)[2]
    v = O.get_value_by_index(qkv16[:, 2 * C:], zero, gs["inds"], S, axis)[2]
    eye = {"p.in_proj_weight": np.concatenate([np.eye(C, dtype=np.float32)] * 3), "p.in_proj_bias": np.zeros(3 * C, np.float32),
           "p.out_proj.weight": np.eye(C, dtype=np.float32), "p.out_proj.bias": np.zeros(C, np.float32)}
    if S == 0:
        return np.zeros((max_pillars, C), np.float32)
    a = D.mha(D.T(q[:S]), D.T(k[:S]), D.T(v[:S]), D.T(gs["mask0_h"][:S]), eye, "p").numpy()       # axis-0 mask for both (:658,:708)
    a_full = np.zeros((gs["inds"].shape[1], 36, C), np.float32); a_full[:S] = a
    return O.map_set_feature2voxel(a_full, gs["inds"], S, axis, max_pillars)


def _run_attention(P, c, qkv16, gs, axis):
    op = P.add_set_attention_op(c["W"], 36, 192, 8, axis, c["P"], io_half=True)          # zero_fill on: rows no set covers are 0
    out = op(dev(qkv16[None]).half(), dev(gs["inds"][None]), dev(gs["mask"][None]), scalar(gs["S"]))[0]
    torch.cuda.synchronize()
    return out[0].float().cpu().numpy()


@pytest.mark.parametrize("axis", [0, 1])
@pytest.mark.parametrize("win", [0, 1])
def test_set_attention_f16_reference_frame(pkg, oracle, axis, win):
    """frame 000000: 454 / 272 sets, two thirds of the slots are masked duplicates"""
    P, O = pkg.plugin, oracle
    c = cases.caps("ref")
    pts, n = cases.load_frame("000000", c["N"])
    vox = O.points2features(pts, n, cases.p2f_cfg(c))
    rw = O.window_partition(vox["coords"], vox["P"], cases.wp_cfg(c, win))
    gs = O.get_set(rw["gidx"], rw["cinw"], rw["vcnt"], rw["W"], cases.gs_cfg(c, win))
    assert (gs["mask"][0, :gs["S"]] < 0).mean() > 0.3
    rng = np.random.default_rng(100 + 2 * win + axis)
    qkv = np.zeros((c["P"], 576), np.float32)
    qkv[:vox["P"]] = rng.standard_normal((vox["P"], 576)) * np.array([0.6] * 192 + [1.5] * 192 + [1.0] * 192, np.float32)
    qkv16 = r16(qkv)
    ref = _attention_reference(O, qkv16, gs, axis, c["P"])
    got = _run_attention(P, c, qkv16, gs, axis)
    Pn = vox["P"]
    # P (softmax) is rounded to fp16 before P V and the result is stored as fp16: 2 x 2^-11 relative
    err = np.abs(got[:Pn] - ref[:Pn])
    assert err.max() < 1.5e-3 * max(1.0, np.abs(ref).max()), err.max()
    assert err.mean() < 1.5e-4
    assert not got[Pn:].any()


def _synthetic_sets(O, c, coords_yx):
    """windows/sets of a hand-made list of pillar cells"""
    coords = np.zeros((c["P"], 4), np.uint32)
    yx = np.array(sorted(coords_yx, key=lambda t: t[0] * 468 + t[1]), np.uint32).reshape(-1, 2)
    coords[:len(yx), 2] = yx[:, 0]; coords[:len(yx), 3] = yx[:, 1]
    rw = O.window_partition(coords, len(yx), cases.wp_cfg(c, 0))
    return len(yx), O.get_set(rw["gidx"], rw["cinw"], rw["vcnt"], rw["W"], cases.gs_cfg(c, 0))


@pytest.mark.parametrize("case", ["empty", "one_voxel", "five_voxels", "full_window", "37_voxels"])
def test_set_attention_f16_small_sets(pkg, oracle, case):
    """S = 0; S = 1 with 35 / 31 duplicate slots; a full 12x12 window (4 sets, no duplicates); 37 voxels (2 sets, 35 duplicates)"""
    P, O = pkg.plugin, oracle
    c = cases.caps("ref")
    cells = {"empty": [], "one_voxel": [(100, 200)], "five_voxels": [(24, 24), (24, 25), (25, 24), (30, 35), (35, 30)],
             "full_window": [(120 + y, 240 + x) for y in range(12) for x in range(12)],
             "37_voxels": [(120 + i // 12, 240 + i % 12) for i in range(37)]}[case]
    Pn, gs = _synthetic_sets(O, c, cells)
    assert gs["S"] == {"empty": 0, "one_voxel": 1, "five_voxels": 1, "full_window": 4, "37_voxels": 2}[case]
    rng = np.random.default_rng(len(cells))
    qkv = np.zeros((c["P"], 576), np.float32); qkv[:Pn] = rng.standard_normal((Pn, 576))
    qkv16 = r16(qkv)
    for axis in (0, 1):
        ref = _attention_reference(O, qkv16, gs, axis, c["P"])
        got = _run_attention(P, c, qkv16, gs, axis)
        assert np.abs(got - ref).max() < 1.5e-3 * max(1.0, np.abs(ref).max())
        if case == "one_voxel":      # softmax over one live key = that voxel's V row, exactly
            assert np.array_equal(got[0], qkv16[0, 384:])


def test_set_attention_f16_set_cap_overflow_leaves_zeros(pkg, oracle):
    """More sets than the cap: GetSet truncates (getSet.cu:147,242 caps S at MAX_WIN_NUM); the attention output rows of the
    dropped voxels must be zero (what the reference's memset gives, mapSetFeature2voxel.cu:314), not stale data."""
    P, O = pkg.plugin, oracle
    c = dict(cases.caps("ref")); c["W"] = 16
    cells = [(12 * wy + 3, 12 * wx + 5) for wy in range(5) for wx in range(6)]            # 30 windows of one voxel each
    Pn, gs = _synthetic_sets(O, c, cells)
    assert gs["S"] == 16
    rng = np.random.default_rng(3)
    qkv16 = r16(np.concatenate([rng.standard_normal((Pn, 576)), np.zeros((c["P"] - Pn, 576))]).astype(np.float32))
    op = P.add_set_attention_op(c["W"], 36, 192, 8, 0, c["P"], io_half=True).set_zero_fill(False)
    args = (dev(qkv16[None]).half(), dev(gs["inds"][None]), dev(gs["mask"][None]), scalar(gs["S"]))
    out = op(*args)[0]
    out.fill_(7.0)                                         # stale data of an earlier frame
    out = op(*args)[0]
    torch.cuda.synchronize()
    ref = _attention_reference(O, qkv16, gs, 0, c["P"])
    got = out[0].float().cpu().numpy()
    covered = np.zeros(c["P"], bool); covered[gs["inds"][0, :16].reshape(-1)] = True
    assert np.abs(got[covered] - ref[covered]).max() < 1.5e-3 * max(1.0, np.abs(ref).max())


# =====================================================================================================================
# encoder_mlp_stream_kernel (DsvtEncoderMlpPlugin) vs the reference wiring src/dsvt-ai-trt.cpp:669-756
# =====================================================================================================================
def _mlp_reference(att16, x, xb, w, lp, block, n, mimic_roundings=True, round_weights=True, rows=None):
    """fp64 restatement of  s1 = LN1(att Wo^T + bo + x); h = GELU(s1 W1^T + b1); y = LN3(LN2(s1 + h W2^T + b2) + x)
    (; y = LN4(y + xb)) with eps = 0 LayerNorms (layerNorm.cu:261-402 semantics: biased variance) and the tanh GELU of
    gelu.cu:201-250, on fp16-rounded weights; mimic_roundings: the two operand roundings the kernel performs (s1 and h are
    MFMA operands of the next GEMM) are restated.  rows: evaluate these rows only (cases.sample_rows; every row is independent)."""
    f = lambda k: w[lp + k].astype(np.float64)
    h16 = (lambda k: r16(w[lp + k]).astype(np.float64)) if round_weights else f

    def ln(v, name):
        g, b = (w[name + ".weight"].astype(np.float64), w[name + ".bias"].astype(np.float64))
        mu = v.mean(1, keepdims=True); var = ((v - mu) ** 2).mean(1, keepdims=True)
        return (v - mu) / np.sqrt(var) * g + b

    sel = slice(0, n) if rows is None else rows
    a = att16[sel].astype(np.float64); x_ = x[sel].astype(np.float64)
    s1 = ln(a @ h16(".win_attn.self_attn.out_proj.weight").T + f(".win_attn.self_attn.out_proj.bias") + x_, lp + ".win_attn.norm1")
    s1_op = r16(s1).astype(np.float64) if mimic_roundings else s1
    u = s1_op @ h16(".win_attn.linear1.weight").T + f(".win_attn.linear1.bias")
    h = 0.5 * u * (1.0 + np.tanh(0.7978845608028654 * (u + 0.044715 * u ** 3)))
    h_op = r16(h).astype(np.float64) if mimic_roundings else h
    s2 = ln(s1 + h_op @ h16(".win_attn.linear2.weight").T + f(".win_attn.linear2.bias"), lp + ".win_attn.norm2")
    y = ln(s2 + x_, lp + ".norm")
    if block is not None:
        y = ln(y + xb[sel].astype(np.float64), f"module.backbone_3d.residual_norm_stage_0.{block}")
    return y


@pytest.mark.parametrize("block_ln,MR,n", [(False, 8192, 5504), (True, 8192, 5504), (False, 8192, 1), (True, 8192, 17),
                                             (True, 65536, 34483),      # nine live waves per workgroup
                                             (False, 65536, 39000),     # ten
                                             (True, 65536, 50000),      # eight-wave workgroups in rounds
                                             (True, 131072, 103449),    # capacity of three frames: four waves x 32 rows, two workgroups per CU
                                             (False, 131072, 33000),    # ... with few rows: the elastic kernel takes them, the other one returns at once
                                             (False, 196608, 137932),   # four frames
                                             (True, 196608, 81921),     # the first row count of the four-wave kernel's regime (threshold 2.5 x 128 x 256)
                                             (False, 196608, 81920),    # ... and the last of the elastic one's
                                             (True, 196608, 131072),    # exactly two rounds of two workgroups per CU
                                             (False, 262144, 131105)])  # ... and 33 rows more: one wave of a third round
def test_encoder_mlp_f16_against_reference_wiring(pkg, oracle, block_ln, MR, n):
    P = pkg.plugin
    rng = np.random.default_rng(7 * n + block_ln)
    C = 192
    w = pkg.synth.make_weights(with_bev=False)
    b_ = 1
    lp = f"module.backbone_3d.stage_0.{b_}.encoder_list.1"
    ln = lambda k: (w[k + ".weight"], w[k + ".bias"])
    lns = [ln(lp + ".win_attn.norm1"), ln(lp + ".win_attn.norm2"), ln(lp + ".norm")]
    if block_ln:
        lns.append(ln(f"module.backbone_3d.residual_norm_stage_0.{b_}"))
    att16 = np.zeros((MR, C), np.float32); att16[:n] = r16(rng.standard_normal((n, C)))
    x = np.zeros((MR, C), np.float32); x[:n] = rng.standard_normal((n, C))
    xb = np.zeros((MR, C), np.float32); xb[:n] = rng.standard_normal((n, C))
    mlp = P.add_encoder_mlp_op(w[lp + ".win_attn.self_attn.out_proj.weight"], w[lp + ".win_attn.self_attn.out_proj.bias"],
                               w[lp + ".win_attn.linear1.weight"], w[lp + ".win_attn.linear1.bias"],
                               w[lp + ".win_attn.linear2.weight"], w[lp + ".win_attn.linear2.bias"], lns, MR)
    args = [dev(att16[None]).half(), scalar(n), dev(x[None])] + ([dev(xb[None])] if block_ln else [])
    got, got_h = mlp(*args)
    torch.cuda.synchronize()
    g = host(got)[0]
    rows = cases.sample_rows(n)                                      # large cases: the fp64 reference on a tile-covering sample of the rows
    sel = slice(0, n) if rows is None else rows
    ref = _mlp_reference(att16, x, xb, w, lp, b_ if block_ln else None, n, rows=rows)
    assert np.isfinite(g[:n]).all()
    err = np.abs(g[sel] - ref)
    # one fp16 ulp flip of an operand element (the kernel rounds an fp32-accumulated value, the reference an fp64 one) moves a
    # LayerNorm-ed O(1) output by ~3e-5; everything else is fp32 summation order
    assert err.max() < 5e-4, err.max()
    assert err.mean() < 2e-5, err.mean()
    # against the arithmetic WITHOUT the internal operand roundings: fp16-sized
    ref0 = _mlp_reference(att16, x, xb, w, lp, b_ if block_ln else None, n, mimic_roundings=False, rows=rows)
    assert np.abs(g[sel] - ref0).max() < 4e-3 and np.abs(g[sel] - ref0).mean() < 3e-4
    assert not g[n:].any()
    gh = got_h[0, :n].float().cpu().numpy()
    assert np.abs(gh - g[:n]).max() <= 2.0 ** -11 * np.abs(g[:n]).max() * 1.01           # the fp16 copy is the same value, rounded


# =====================================================================================================================
# pfn_kernel (DsvtPillarFeatureNetPlugin) vs the oracle's PFN: FC+BN+ReLU -> TorchScatterMax -> concat -> FC+BN+ReLU -> max
# =====================================================================================================================
@pytest.mark.parametrize("frame,capname,n_pts", [("000000", "ref", 0), ("000004", "ref", 0), (None, "mid", 60000)])
def test_pillar_feature_net_against_oracle(pkg, oracle, frame, capname, n_pts):
    """dense_ref.voxel_stage = src/dsvt-ai-trt.cpp:571-589 with the C oracle's TorchScatterMax.  Layer 0 of the fused kernel is
    fp32 MFMA (exact products); layer 1 runs on fp16 operands (x0, max(x0) and W1 rounded to fp16): 2^-11 relative per operand
    over a 192-term dot product => ~1e-3 of the feature scale."""
    from oracle import dense_ref as D
    P, O = pkg.plugin, oracle
    c = cases.caps(capname)
    if frame:
        pts, n = cases.load_frame(frame, c["N"])
    else:
        pts, n = cases.pad_points(pkg.synth.lidar_like(n_pts, 0), c["N"])
    w = pkg.synth.make_weights(with_bev=False)
    cfg = D.OracleCfg(max_points=c["N"], max_points_filter=c["Nk"], max_pillars=c["P"], max_win=c["W"], blocks=0)
    ost = D.voxel_stage(pts, n, w, cfg)
    W0, b0 = pkg.pipeline.fold_linear_bn(w, "module.vfe.pfn_layers.0.linear", "module.vfe.pfn_layers.0.norm", 1e-5)
    W1, b1 = pkg.pipeline.fold_linear_bn(w, "module.vfe.pfn_layers.1.linear", "module.vfe.pfn_layers.1.norm", 1e-5)
    feat, pidx, coords, pcnt, Pn, Nk = make_voxelizer(P, c)(dev(pts[None]), scalar(n))
    v, v16 = P.add_pillar_feature_net_op(c["P"], W0, b0, W1, b1)(feat, pidx, pcnt, Pn)
    torch.cuda.synchronize()
    np_ = ost["P"]
    assert int(Pn.cpu()[0]) == np_
    got, ref = host(v)[0], ost["vfeat"]
    scale = np.abs(ref[:np_]).max()
    err = np.abs(got[:np_] - ref[:np_])
    assert err.max() < 1.5e-3 * scale, (err.max(), scale)
    assert err.mean() < 1.5e-4 * scale
    assert not got[np_:].any()
    assert np.abs(v16[0, :np_].float().cpu().numpy() - got[:np_]).max() <= 2.0 ** -11 * scale * 1.01
    # packing pillars with <= 4 points four to an MFMA tile does not change a bit: a tile row depends on its own point only
    v1, _ = P.add_pillar_feature_net_op(c["P"], W0, b0, W1, b1, pack_small_pillars=False)(feat, pidx, pcnt, Pn)
    torch.cuda.synchronize()
    assert torch.equal(v, v1)
    cnts = host(pcnt)[0, :np_, 0]
    assert (cnts <= 4).mean() > 0.4 and (cnts > 16).any()                    # both tile kinds are exercised


# =====================================================================================================================
# linear_f16_rows_kernel (the QKV projection: DsvtLinearPlugin, N = 576, add_cols = 384, fp16 in / out)
# =====================================================================================================================
@pytest.mark.parametrize("MR,n", [(65536, 34483), (65536, 39000), (65536, 50000), (8192, 100), (8192, 5504), (8192, 1),
                                  # row capacity of three or more frames: linear_f16_resident_kernel (weights resident in LDS, waves walk the tiles)
                                  (262144, 137932), (262144, 100), (196608, 190001), (262144, 1)])
@pytest.mark.parametrize("table", [False, True])
def test_qkv_rows_kernel_against_fp64_product(pkg, MR, n, table):
    """q = k = (x + pos) Wqk^T + b, v = x Wv^T + b (getValueByIndex.cu:282-355 + src/dsvt-ai-trt.cpp:328-330, per voxel row) on
    fp16 operands, every row regime (8 / 9 / 10 live waves, rounds), with the position rows as a tensor or gathered from the
    per-window cell table; fp64 product of the same operands.  Output is stored as fp16: 2^-11 of the element, bound 1e-3 of scale."""
    P = pkg.plugin
    g = torch.Generator(device="cpu").manual_seed(MR + n)
    C, wx = 192, 12
    x = torch.randn((1, MR, C), generator=g).half()
    W = (torch.randn((3 * C, C), generator=g) / np.sqrt(C)); b = torch.randn(3 * C, generator=g) * 0.1
    kw = dict(add_cols=2 * C, compute_type=P.COMPUTE_F16, input_half=True, output_mode=P.OUT_F16)
    cnt = torch.tensor([n], dtype=torch.int32, device=DEV)
    if table:
        tab = (torch.randn((1, wx * wx, C), generator=g) * 0.5).half()
        c2d = torch.zeros((1, MR, 3), dtype=torch.int32)
        c2d[0, :, 1] = torch.randint(0, wx, (MR,), generator=g); c2d[0, :, 2] = torch.randint(0, wx, (MR,), generator=g)
        pos = tab[0][(c2d[0, :, 1] * wx + c2d[0, :, 2]).long()][None]
        got = P.add_linear_op(W.numpy(), b.numpy(), MR, add_gather_width=wx, **kw)(x.to(DEV), cnt, tab.to(DEV), c2d.to(DEV))[0]
    else:
        pos = (torch.randn((1, MR, C), generator=g) * 0.5).half()
        got = P.add_linear_op(W.numpy(), b.numpy(), MR, **kw)(x.to(DEV), cnt, pos.to(DEV))[0]
    torch.cuda.synchronize()
    Wd = W.half().double()
    xs = (x[0, :n].float() + pos[0, :n].float()).half().double()          # the A prologue adds in fp16 (one rounding)
    ref = torch.cat([xs @ Wd[:2 * C].T, x[0, :n].double() @ Wd[2 * C:].T], 1) + b.double()
    err = (got[0, :n].double().cpu() - ref).abs()
    scale = ref.abs().max().item()
    assert err.max().item() < 1e-3 * scale, (err.max().item(), scale)
    assert err.mean().item() < 1e-4 * scale
    assert not got[0, n:].any()


def test_encoder_mlp_frames_field_picks_the_kernel_not_the_result(pkg):
    """optional field `frames` of DsvtEncoderMlpPlugin: 1-2 = the elastic kernel, >= 3 = four waves x 32 rows (two workgroups per CU), absent =
    decided by the row count on the device.  Either kernel is correct for any count and both sum a row's products in the same k order: same
    bits from all three, before and after a serialise / deserialise round trip (the field travels as a trailing int)."""
    P = pkg.plugin
    rng = np.random.default_rng(11)
    C, MR, n = 192, 131072, 70001
    w = pkg.synth.make_weights(with_bev=False)
    lp = "module.backbone_3d.stage_0.0.encoder_list.0"
    ln = lambda k: (w[k + ".weight"], w[k + ".bias"])
    lns = [ln(lp + ".win_attn.norm1"), ln(lp + ".win_attn.norm2"), ln(lp + ".norm")]
    att = torch.from_numpy(r16(rng.standard_normal((1, MR, C)))).half().to(DEV)
    x = torch.from_numpy(rng.standard_normal((1, MR, C)).astype(np.float32)).to(DEV)
    cnt = scalar(n)
    mk = lambda fr: P.add_encoder_mlp_op(w[lp + ".win_attn.self_attn.out_proj.weight"], w[lp + ".win_attn.self_attn.out_proj.bias"],
                                         w[lp + ".win_attn.linear1.weight"], w[lp + ".win_attn.linear1.bias"],
                                         w[lp + ".win_attn.linear2.weight"], w[lp + ".win_attn.linear2.bias"], lns, MR, frames=fr)
    outs = {}
    for fr in (0, 2, 4):
        op = mk(fr)
        outs[fr] = [t.clone() for t in op(att, cnt, x)]
        blob = op.serialize()
        again = P.Plugin.deserialize("DsvtEncoderMlpPlugin", blob)
        assert again.serialize() == blob
        o2 = again(att, cnt, x)
        torch.cuda.synchronize()
        assert torch.equal(o2[0], outs[fr][0]) and torch.equal(o2[1], outs[fr][1])
    assert len(mk(4).serialize()) == len(mk(0).serialize()) + 4
    for fr in (2, 4):
        assert torch.equal(outs[fr][0][0, :n], outs[0][0][0, :n]) and torch.equal(outs[fr][1][0, :n], outs[0][1][0, :n])
    assert float(outs[0][0][0, :n].abs().max()) > 0.1


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_qkv_resident_kernel_row_counts(pkg, seed):
    """linear_f16_resident_kernel (row capacity of four frames) on random row counts -- fewer rows than waves, a ragged last tile, the
    capacity itself -- with the position rows gathered through the window-cell table"""
    P = pkg.plugin
    rng = np.random.default_rng(100 + seed)
    MR, C, wx = 262144, 192, 24
    n = int([1, 15, 16 * 8 * 128 + 3, rng.integers(2, MR), rng.integers(2, MR), MR][seed])
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn((1, MR, C), generator=g).half()
    W = (torch.randn((3 * C, C), generator=g) / np.sqrt(C)); b = torch.randn(3 * C, generator=g) * 0.1
    tab = (torch.randn((1, wx * wx, C), generator=g) * 0.5).half()
    c2d = torch.zeros((1, MR, 3), dtype=torch.int32)
    c2d[0, :, 1] = torch.randint(0, wx, (MR,), generator=g); c2d[0, :, 2] = torch.randint(0, wx, (MR,), generator=g)
    op = P.add_linear_op(W.numpy(), b.numpy(), MR, add_cols=2 * C, compute_type=P.COMPUTE_F16, input_half=True, output_mode=P.OUT_F16, add_gather_width=wx)
    got = op(x.to(DEV), torch.tensor([n], dtype=torch.int32, device=DEV), tab.to(DEV), c2d.to(DEV))[0]
    torch.cuda.synchronize()
    Wd = W.half().double().to(DEV)
    xd = x[0, :n].to(DEV); pos = tab[0].to(DEV)[(c2d[0, :n, 1] * wx + c2d[0, :n, 2]).long().to(DEV)]
    xs = (xd.float() + pos.float()).half().double()
    ref = torch.cat([xs @ Wd[:2 * C].T, xd.double() @ Wd[2 * C:].T], 1) + b.double().to(DEV)
    err = (got[0, :n].double() - ref).abs()
    assert err.max().item() < 1e-3 * ref.abs().max().item()
    assert not got[0, n:].any()
