"""Kernel-level parity of the split-precision ("fp32 grade on the fp16 matrix cores") kernels of round 3 -- linear_split_rows_kernel,
encoder_mlp_stream_kernel<.., SPLIT>, pfn_kernel<SPLIT> -- against float64 restatements of the reference's fp32 wiring on the SAME fp32
operands, and of the frame pipeline built from them against the fp32 oracle at the north-star tolerance (boxes within 1e-3).

Every operand is the pair hi = fp16(v), lo = fp16(v - hi); a product is w_hi a_hi + w_lo a_hi + w_hi a_lo with fp32 accumulation: the
dropped term is 2^-22 of the product, so these kernels must sit at fp32-summation-order distance (1e-6-class) from the fp64 result,
three orders below the fp16 kernels' bounds in tests/test_f16_kernels_gpu.py.
"""
import numpy as np
import pytest
import torch

from tests import cases
from tests.test_plugins_gpu import dev, host, scalar, make_voxelizer
from tests.test_f16_kernels_gpu import _mlp_reference

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("MR,n", [(8192, 1), (8192, 100), (8192, 5504), (65536, 34483), (262144, 137932), (196608, 190001)])
@pytest.mark.parametrize("table", [False, True])
def test_qkv_split_kernel_against_fp64_product(pkg, MR, n, table):
    """q = k = (x + pos) Wqk^T + b, v = x Wv^T + b (getValueByIndex.cu:282-355 + src/dsvt-ai-trt.cpp:328-330, per voxel row), fp32 tensors,
    split-precision products: within 2e-6 of scale of the fp64 product of the fp32 operands (the fp16 kernel's bound is 1e-3)."""
    P = pkg.plugin
    g = torch.Generator(device="cpu").manual_seed(MR + n)
    C, wx = 192, 12
    x = torch.randn((1, MR, C), generator=g)
    W = (torch.randn((3 * C, C), generator=g) / np.sqrt(C)); b = torch.randn(3 * C, generator=g) * 0.1
    kw = dict(add_cols=2 * C, compute_type=P.COMPUTE_SPLIT)
    cnt = torch.tensor([n], dtype=torch.int32, device=DEV)
    if table:
        tab = torch.randn((1, wx * wx, C), generator=g) * 0.5
        c2d = torch.zeros((1, MR, 3), dtype=torch.int32)
        c2d[0, :, 1] = torch.randint(0, wx, (MR,), generator=g); c2d[0, :, 2] = torch.randint(0, wx, (MR,), generator=g)
        pos = tab[0][(c2d[0, :, 1] * wx + c2d[0, :, 2]).long()][None]
        op = P.add_linear_op(W.numpy(), b.numpy(), MR, add_gather_width=wx, **kw)
        got = op(x.to(DEV), cnt, tab.to(DEV), c2d.to(DEV))[0]
    else:
        pos = torch.randn((1, MR, C), generator=g) * 0.5
        op = P.add_linear_op(W.numpy(), b.numpy(), MR, **kw)
        got = op(x.to(DEV), cnt, pos.to(DEV))[0]
    torch.cuda.synchronize()
    assert got.dtype == torch.float32
    Wd = W.double()
    xs = (x[0, :n] + pos[0, :n]).double()                                  # the prologue adds in fp32
    ref = torch.cat([xs @ Wd[:2 * C].T, x[0, :n].double() @ Wd[2 * C:].T], 1) + b.double()
    err = (got[0, :n].double().cpu() - ref).abs()
    scale = ref.abs().max().item()
    assert err.max().item() < 2e-6 * scale, (err.max().item(), scale)
    assert not got[0, n:].any()
    again = P.Plugin.deserialize("DsvtLinearPlugin", op.serialize())
    assert again.serialize() == op.serialize()


@pytest.mark.parametrize("N,add_cols,n", [(384, 192, 5504), (192, 0, 777), (960, 192, 130), (768, 768, 3001), (576, 0, 34483)])
def test_split_linear_of_other_shapes_on_the_streamed_kernel(pkg, N, add_cols, n):
    """The QKV shape (N = 576, add_cols a multiple of 192) runs on linear_split_resident_kernel; every other split-precision linear -- other widths, or a
    position-added column range that is not whole thirds -- on linear_split_rows_kernel (all column chunks of a 128-row tile per workgroup, weights
    streamed): the same 2e-6 of scale against the fp64 product."""
    P = pkg.plugin
    g = torch.Generator(device="cpu").manual_seed(N + n)
    C, MR = 192, 65536
    x = torch.randn((1, MR, C), generator=g); pos = torch.randn((1, MR, C), generator=g) * 0.5
    W = (torch.randn((N, C), generator=g) / np.sqrt(C)); b = torch.randn(N, generator=g) * 0.1
    op = P.add_linear_op(W.numpy(), b.numpy(), MR, add_cols=add_cols, compute_type=P.COMPUTE_SPLIT)
    cnt = torch.tensor([n], dtype=torch.int32, device=DEV)
    got = (op(x.to(DEV), cnt, pos.to(DEV)) if add_cols else op(x.to(DEV), cnt))[0]
    torch.cuda.synchronize()
    Wd = W.double()
    xs = (x[0, :n] + pos[0, :n]).double()
    ref = torch.cat([xs @ Wd[:add_cols].T, x[0, :n].double() @ Wd[add_cols:].T], 1) + b.double()
    err = (got[0, :n].double().cpu() - ref).abs()
    assert err.max().item() < 2e-6 * ref.abs().max().item()
    assert not got[0, n:].any()


def test_split_operand_range(pkg):
    """operands far from 1: tiny weights (lo parts in the fp16 subnormal range: absolute step 2^-24, i.e. the split carries
    ~2^-18 of a 0.01-sized operand, not 2^-22) and activations beyond the fp16 range (hi saturates at 65504, no inf / NaN)."""
    P = pkg.plugin
    g = torch.Generator(device="cpu").manual_seed(5)
    C, MR, n = 192, 8192, 4000
    x = torch.randn((1, MR, C), generator=g)
    W = torch.randn((3 * C, C), generator=g) * 0.01
    cnt = torch.tensor([n], dtype=torch.int32, device=DEV)
    got = P.add_linear_op(W.numpy(), None, MR, compute_type=P.COMPUTE_SPLIT)(x.to(DEV), cnt)[0]
    ref = x[0, :n].double() @ W.double().T
    err = (got[0, :n].double().cpu() - ref).abs().max().item()
    assert err < 2e-5 * ref.abs().max().item(), err                      # subnormal lo parts: 2^-18-class, still 50x inside 1e-3
    x[0, 0, :4] = torch.tensor([1e5, -2e5, 7e4, 65504.0])
    got = P.add_linear_op(W.numpy(), None, MR, compute_type=P.COMPUTE_SPLIT)(x.to(DEV), cnt)[0]
    torch.cuda.synchronize()
    assert torch.isfinite(got).all()
    # (row 0 is coarse -- its saturated elements carry at most +-131008 -- but finite; every other row is untouched)
    assert (got[0, 1:n].double().cpu() - ref[1:]).abs().max().item() < 2e-5 * ref.abs().max().item()


@pytest.mark.parametrize("block_ln,MR,n,frames", [(False, 8192, 5504, 0), (True, 8192, 5504, 0), (False, 8192, 1, 0), (True, 8192, 17, 0),
                                                    (True, 65536, 34483, 0), (False, 196608, 137932, 0), (True, 196608, 131072 + 33, 0),
                                                    # frames=1: the ten-wave elastic kernel with nine, ten and (two rounds) eight live waves
                                                    (True, 65536, 34483, 1), (False, 65536, 40951, 1), (True, 65536, 47003, 1), (False, 65536, 3, 1)])
def test_encoder_mlp_split_against_reference_wiring(pkg, block_ln, MR, n, frames):
    """src/dsvt-ai-trt.cpp:669-756 in fp64 on the fp32 operands, NO operand rounding anywhere: the split kernel must be within
    fp32-arithmetic distance (LayerNorm outputs are O(1): 2e-5 max, 1e-6 mean), 200x below the fp16 kernel's distance."""
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
    att = np.zeros((MR, C), np.float32); att[:n] = rng.standard_normal((n, C))
    x = np.zeros((MR, C), np.float32); x[:n] = rng.standard_normal((n, C))
    xb = np.zeros((MR, C), np.float32); xb[:n] = rng.standard_normal((n, C))
    mlp = P.add_encoder_mlp_op(w[lp + ".win_attn.self_attn.out_proj.weight"], w[lp + ".win_attn.self_attn.out_proj.bias"],
                               w[lp + ".win_attn.linear1.weight"], w[lp + ".win_attn.linear1.bias"],
                               w[lp + ".win_attn.linear2.weight"], w[lp + ".win_attn.linear2.bias"], lns, MR, split_precision=True, frames=frames)
    assert mlp.nb_outputs == 1
    args = [dev(att[None]), scalar(n), dev(x[None])] + ([dev(xb[None])] if block_ln else [])
    got, = mlp(*args)
    torch.cuda.synchronize()
    g = host(got)[0]
    rows = cases.sample_rows(n)                                      # large cases: the fp64 reference on a tile-covering sample of the rows
    ref = _mlp_reference(att, x, xb, w, lp, b_ if block_ln else None, n, mimic_roundings=False, round_weights=False, rows=rows)
    assert np.isfinite(g[:n]).all()
    err = np.abs(g[slice(0, n) if rows is None else rows] - ref)
    assert err.max() < 2e-5, err.max()
    assert err.mean() < 1.5e-6, err.mean()
    assert not g[n:].any()
    blob = mlp.serialize()
    again = P.Plugin.deserialize("DsvtEncoderMlpPlugin", blob)
    assert again.serialize() == blob and again.nb_outputs == 1
    assert torch.equal(again(*args)[0], got)


@pytest.mark.parametrize("frame,capname,n_pts", [("000000", "ref", 0), (None, "mid", 60000)])
def test_pillar_feature_net_split_against_oracle(pkg, oracle, frame, capname, n_pts):
    """dense_ref.voxel_stage (src/dsvt-ai-trt.cpp:571-589, fp32) against the split-precision pillar feature net: 1e-5 of the feature
    scale (fp32 summation order), where the fp16 variant sits at 1.5e-3."""
    from oracle import dense_ref as D
    P = pkg.plugin
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
    op = P.add_pillar_feature_net_op(c["P"], W0, b0, W1, b1, split_precision=True)
    v, = op(feat, pidx, pcnt, Pn)
    torch.cuda.synchronize()
    np_ = ost["P"]
    got, ref = host(v)[0], ost["vfeat"]
    scale = np.abs(ref[:np_]).max()
    err = np.abs(got[:np_] - ref[:np_])
    assert err.max() < 1e-5 * scale, (err.max(), scale)
    assert not got[np_:].any()
    again = P.Plugin.deserialize("DsvtPillarFeatureNetPlugin", op.serialize())
    assert again.serialize() == op.serialize() and torch.equal(again(feat, pidx, pcnt, Pn)[0], v)


def _pfn_fp64(feat, counts, W0, b0, W1, b1):
    """src/dsvt-ai-trt.cpp:565-589 in fp64 on the fp32 operands: x0 = ReLU(FC0 f), m = max over the pillar, x1 = ReLU(FC1 [x0 | m]), max over the pillar"""
    f = feat.astype(np.float64)
    x0 = np.maximum(f @ W0.astype(np.float64).T + b0, 0.0)
    starts = np.concatenate([[0], np.cumsum(counts)])
    seg = np.repeat(np.arange(len(counts)), counts)
    m = np.maximum.reduceat(x0, starts[:-1], axis=0)
    x1 = np.maximum(np.concatenate([x0, m[seg]], axis=1) @ W1.astype(np.float64).T + b1, 0.0)
    return np.maximum.reduceat(x1, starts[:-1], axis=0)


@pytest.mark.parametrize("split", [True, False])
@pytest.mark.parametrize("kind,P_", [("ones", 1), ("ones", 17), ("ones", 3000), ("full", 1), ("full", 16), ("full", 700), ("mixed", 5), ("mixed", 1023), ("mixed", 1025),
                                      ("mixed", 20000), ("mixed", 60000), ("lumpy", 50000), ("huge", 9)])
def test_pillar_feature_net_any_pillar_sizes(pkg, split, kind, P_):
    """pfn_kernel cuts the pillars into groups of equal WORK (8 + points per pillar, 128 units per group) and finds a group's first pillar by searching the
    row-start prefix: tables of every shape -- single points only (sixteen pillars per group), full 48-point pillars (two or three per group), runs of
    dense pillars between sparse ones (the coarse samples of the search fall into very unequal intervals), launches on both sides of the ~47k pillars
    where the search adds its probe round, one pillar, and pillars of more points than a group's budget (200: their groups have no other member and the
    intervals between them hold no pillar at all).  fp64 restatement of the reference wiring on the same fp32 rows; 1e-5 of the scale in split precision,
    1.5e-3 with fp16 operands."""
    P = pkg.plugin
    rng = np.random.default_rng(1000 * P_ + len(kind))
    T = 48
    if kind == "ones": counts = np.ones(P_, np.int64)
    elif kind == "full": counts = np.full(P_, T, np.int64)
    elif kind == "huge": counts = np.where(np.arange(P_) % 3 == 1, 200, rng.integers(1, 6, P_)).astype(np.int64)
    elif kind == "lumpy":
        counts = rng.integers(1, 3, P_).astype(np.int64)
        for a_ in rng.integers(0, P_ - 400, 12): counts[a_:a_ + rng.integers(50, 400)] = rng.integers(30, T + 1)
    else: counts = np.minimum(rng.geometric(0.25, P_), T).astype(np.int64)
    Tcap = int(max(T, counts.max()))
    Nk = int(counts.sum()); cap_p = P_ + 37; cap_n = Nk + 11
    w = pkg.synth.make_weights(with_bev=False)
    W0, b0 = pkg.pipeline.fold_linear_bn(w, "module.vfe.pfn_layers.0.linear", "module.vfe.pfn_layers.0.norm", 1e-5)
    W1, b1 = pkg.pipeline.fold_linear_bn(w, "module.vfe.pfn_layers.1.linear", "module.vfe.pfn_layers.1.norm", 1e-5)
    feat = np.zeros((1, cap_n, 10), np.float32)
    feat[0, :Nk] = rng.standard_normal((Nk, 10)).astype(np.float32) * np.array([30, 30, 2, 0.3, 0.2, 0.2, 0.5, 0.1, 0.1, 1.0], np.float32)
    starts = np.concatenate([[0], np.cumsum(counts)])[:-1]
    pidx = np.zeros((1, cap_p, Tcap), np.int32); pcnt = np.zeros((1, cap_p, 1), np.int32)
    pidx[0, :P_, 0] = starts                                       # (the fused kernel reads slot 0 = the pillar's first row; the rows of a pillar are consecutive)
    pcnt[0, :P_, 0] = counts
    op = P.add_pillar_feature_net_op(cap_p, W0, b0, W1, b1, split_precision=split)
    out = op(dev(feat), dev(pidx), dev(pcnt), scalar(P_))
    torch.cuda.synchronize()
    got = host(out[0])[0]
    ref = _pfn_fp64(feat[0, :Nk], counts, W0, b0, W1, b1)
    scale = np.abs(ref).max()
    err = np.abs(got[:P_] - ref).max()
    assert err < (1e-5 if split else 1.5e-3) * scale, (err, scale)
    assert not got[P_:].any()
    if not split:
        assert np.abs(host(out[1])[0, :P_].astype(np.float32) - got[:P_]).max() <= 2.0 ** -11 * scale * 1.01


@pytest.mark.parametrize("frame", ["000000", "lidar180000"])
def test_backbone_features_split_mode(pkg, oracle, frame):
    """voxel features after the four DSVT blocks, split-precision frame path against the fp32 oracle (2e-4: the bar of the exact-fp32 mode)"""
    from oracle import dense_ref as D
    from tests.test_pipeline_gpu import _frame_and_caps, _oracle_cfg
    w = pkg.synth.make_weights()
    caps, pts, n = _frame_and_caps(pkg, frame)
    pipe = pkg.pipeline.DsvtPipeline(w, caps=caps, device=DEV, linear_compute=pkg.plugin.COMPUTE_SPLIT, with_head=False)
    x, st = pipe.forward(torch.from_numpy(pts[None]).to(DEV), torch.tensor([n], dtype=torch.int32, device=DEV))
    torch.cuda.synchronize()
    cfg = _oracle_cfg(caps)
    ost = D.voxel_stage(pts, n, w, cfg)
    ref = D.dsvt_blocks(ost, w, cfg)
    np_ = ost["P"]
    err = np.abs(x[0, :np_].cpu().numpy() - np.asarray(ref)[:np_]).max()
    assert err < 2e-4, err


@pytest.mark.parametrize("frame", ["000000", "000004", "lidar180000", "lidar60000s3"])
def test_boxes_split_mode(pkg, oracle, frame):
    """the split-precision frame (bench.py's headline mode) against the fp32 oracle at the north-star tolerance, all nine columns"""
    from tests import golden_oracle as GO
    from tests.parity import match_boxes
    from tests.test_pipeline_gpu import _frame_and_caps, _run
    w = pkg.synth.make_weights()
    caps, pts, n = _frame_and_caps(pkg, frame)
    pipe = pkg.pipeline.DsvtPipeline(w, caps=caps, device=DEV, linear_compute=pkg.plugin.COMPUTE_SPLIT)
    boxes, cnt = _run(pkg, pipe, pts, n)
    torch.cuda.synchronize()
    eb, ec = GO.forward(frame, pts, n, w, caps)
    worst, unmatched = match_boxes(boxes[0].cpu().numpy(), int(cnt[0]), eb, ec)
    print("split-mode boxes", frame, "max|diff|", worst, "unmatched", unmatched, "count", int(cnt[0]), ec)
    assert unmatched == 0 and worst < 1e-3, (worst, unmatched)


# ---------------------------------------------------------------------------------------------------------------------
# The configuration bench.py TIMES, under test: the headline mode x four frames per forward x HIP-graph replay x two
# pipelines on two streams, and the whole box row -- all nine columns, yaw included -- against the fp32 oracle on clouds that include the
# ill-conditioned one of the 24-cloud sweep (seed 21: a box whose rot vector is ~1 % of the head's scale, profiles/r04_mx_box_sweep.txt).
# ---------------------------------------------------------------------------------------------------------------------
_ORACLE_BOXES = {}
_WEIGHTS = []


def _weights(pkg):
    if not _WEIGHTS:
        _WEIGHTS.append(pkg.synth.make_weights())
    return _WEIGHTS[0]


def _oracle_boxes(pkg, seed):
    """FilterBoxByScore rows of lidar_like(180000, seed) on the fp32 CPU oracle: the committed fixture (tests/golden_oracle.py; ~10 s each when run live)"""
    if seed not in _ORACLE_BOXES:
        from tests import golden_oracle as GO
        caps = pkg.pipeline.Caps()
        pts, n = cases.pad_points(pkg.synth.lidar_like(180000, seed), caps.N)
        _ORACLE_BOXES[seed] = GO.forward(f"lidar180000s{seed}", pts, n, _weights(pkg), caps)
    return _ORACLE_BOXES[seed]


def test_boxes_split_mode_four_frames_graph(pkg, oracle):
    """DsvtPipeline(COMPUTE_SPLIT, frames=4) as bench.py runs it: (a) each frame's FilterBoxByScore rows are the BITS of the frames=1 split pipeline's --
    eager, replayed from a HIP graph, and replayed by two pipelines on two streams at once (the four-frame launches pick other kernel instantiations:
    eight-wave MLP + two-wave tail round, the resident QKV's row streams, image stacks in the convolutions); (b) every one of the nine box columns
    of every frame is within 1e-3 of dense_ref.forward."""
    from tests.parity import match_boxes
    P = pkg.plugin
    w = pkg.synth.make_weights()
    seeds = [21, 1, 9, 3]
    clouds = [pkg.synth.lidar_like(180000, s) for s in seeds]
    one = pkg.pipeline.DsvtPipeline(w, caps=pkg.pipeline.Caps(), device=DEV, linear_compute=P.COMPUTE_SPLIT)
    singles = []
    for p in clouds:
        pts, n = cases.pad_points(p, one.caps.N)
        r, c = one.forward(torch.from_numpy(pts[None]).to(DEV), torch.tensor([n], dtype=torch.int32, device=DEV))
        torch.cuda.synchronize()
        singles.append((r[0].clone(), int(c[0])))
    del one
    caps4 = pkg.pipeline.Caps.for_frames(4)
    streams = [torch.cuda.Stream(device=DEV) for _ in range(2)]
    pipes = [pkg.pipeline.DsvtPipeline(w, caps=caps4, device=DEV, linear_compute=P.COMPUTE_SPLIT, frames=4) for _ in range(2)]
    orders = [(0, 1, 2, 3), (2, 3, 1, 0)]                     # the two pipelines see the clouds in different slots

    def inputs(order):
        buf = np.zeros((1, 4 * caps4.N, 4), np.float32)
        for slot, k in enumerate(order):
            buf[0, slot * caps4.N:slot * caps4.N + clouds[k].shape[0]] = clouds[k]
        return torch.from_numpy(buf).to(DEV), torch.tensor([clouds[k].shape[0] for k in order], dtype=torch.int32, device=DEV)

    def check(rows, cnt, order, what):
        assert rows.shape == (4, 500, 9) and cnt.shape == (4,)
        for slot, k in enumerate(order):
            assert int(cnt[slot]) == singles[k][1] and singles[k][1] > 0, (what, slot)
            assert torch.equal(rows[slot], singles[k][0]), (what, slot, float((rows[slot] - singles[k][0]).abs().max()))

    ins = [inputs(o) for o in orders]
    rows, cnt = pipes[0].forward(*ins[0])                     # eager
    torch.cuda.synchronize()
    check(rows, cnt, orders[0], "eager")
    outs = []
    for s in range(2):                                        # capture on each pipeline's own stream
        with torch.cuda.stream(streams[s]):
            outs.append(pipes[s].capture(*ins[s]))
            torch.cuda.synchronize()
    for rep in range(3):                                      # both graphs in flight at once, three rounds
        for s in range(2):
            with torch.cuda.stream(streams[s]):
                pipes[s].replay()
        torch.cuda.synchronize()
        for s in range(2):
            check(outs[s][0], outs[s][1], orders[s], f"two-stream replay {rep}")
    # (b) all nine columns against the fp32 oracle
    for slot, k in enumerate(orders[0]):
        eb, ec = _oracle_boxes(pkg, seeds[k])
        worst, unmatched = match_boxes(outs[0][0][slot].cpu().numpy(), int(outs[0][1][slot]), eb, ec)
        print("four-frame graph replay, seed", seeds[k], "max|diff| over nine columns", worst, "unmatched", unmatched)
        assert unmatched == 0 and worst < 1e-3, (seeds[k], worst, unmatched)


def test_boxes_split_mode_eight_clouds_all_nine_columns(pkg, oracle):
    """an 8-seed slice of tools/head_variant_sweep.py as a test: the headline mode's rows against the fp32 ORACLE on eight 180k-point clouds, the sweep's
    worst yaw cases among them (seeds 21, 9, 3, 1); the bar is 1e-3 on every column, and HALF of it on the worst cloud is asserted as the margin a
    later kernel tweak may use up (a 1-ulp change elsewhere moved the fp8 head's worst yaw from 1.7e-3 to 2.0e-3)."""
    from tests.parity import match_boxes
    P = pkg.plugin
    w = pkg.synth.make_weights()
    pipe = pkg.pipeline.DsvtPipeline(w, caps=pkg.pipeline.Caps(), device=DEV, linear_compute=P.COMPUTE_SPLIT)
    worst_all = 0.0
    for seed in (21, 9, 3, 1, 0, 7, 16, 23):
        pts, n = cases.pad_points(pkg.synth.lidar_like(180000, seed), pipe.caps.N)
        r, c = pipe.forward(torch.from_numpy(pts[None]).to(DEV), torch.tensor([n], dtype=torch.int32, device=DEV))
        torch.cuda.synchronize()
        eb, ec = _oracle_boxes(pkg, seed)
        worst, unmatched = match_boxes(r[0].cpu().numpy(), int(c[0]), eb, ec)
        print("seed", seed, "max|diff| over nine columns", worst, "unmatched", unmatched)
        assert unmatched == 0 and worst < 1e-3, (seed, worst, unmatched)
        worst_all = max(worst_all, worst)
    assert worst_all < 5e-4, worst_all


@pytest.mark.parametrize("frame,seed", [("000000", None), ("lidar", 0), ("lidar", 21)])
def test_boxes_fp8_head_variant(pkg, oracle, frame, seed):
    """DsvtPipeline(head_mx=True), the opt-in fast variant of the fp32-grade frame (round 4's headline; bench.py `fp8_head_mode`): the two correction
    products of the head convolutions on the fp8 scaled MFMA.  Centres / sizes / scores stay ~1e-4 from the fp32 oracle (asserted at 3e-4); its yaw does
    NOT hold 1e-3 on boxes with a short rot vector (seed 21: 1.7e-3) -- asserted only at 5e-3, which is why the variant is not the default."""
    from tests import golden_oracle as GO
    from tests.test_pipeline_gpu import _frame_and_caps, _run, _box_errors
    w = pkg.synth.make_weights()
    caps, pts, n = _frame_and_caps(pkg, frame if seed is None else f"lidar180000s{seed}")
    pipe = pkg.pipeline.DsvtPipeline(w, caps=caps, device=DEV, linear_compute=pkg.plugin.COMPUTE_SPLIT, head_mx=True)
    assert pipe.head_mx
    boxes, cnt = _run(pkg, pipe, pts, n)
    torch.cuda.synchronize()
    eb, ec = _oracle_boxes(pkg, seed) if seed is not None else GO.forward(frame, pts, n, w, caps)
    err, frac = _box_errors(boxes[0].cpu().numpy(), int(cnt[0]), eb, ec)
    print("fp8-head box errors per field", frame, seed, err, "matched", frac)
    assert frac == 1.0 and int(cnt[0]) == ec
    assert max(err[:6].max(), err[8]) < 3e-4 and err[6] < 5e-3, err


# =====================================================================================================================
# set_attention_split_kernel (DsvtSetAttentionPlugin split_precision) vs GetValueByIndex -> multHeadAttention core -> MapSetFeature2Voxel
# =====================================================================================================================
def _run_attention_split(P, c, qkv, gs, axis):
    op = P.add_set_attention_op(c["W"], 36, 192, 8, axis, c["P"], split_precision=True)
    out = op(dev(qkv[None]), dev(gs["inds"][None]), dev(gs["mask"][None]), scalar(gs["S"]))[0]
    torch.cuda.synchronize()
    again = P.Plugin.deserialize("DsvtSetAttentionPlugin", op.serialize())
    assert torch.equal(again(dev(qkv[None]), dev(gs["inds"][None]), dev(gs["mask"][None]), scalar(gs["S"]))[0], out)
    return out[0].cpu().numpy()


@pytest.mark.parametrize("axis", [0, 1])
@pytest.mark.parametrize("win", [0, 1])
def test_set_attention_split_reference_frame(pkg, oracle, axis, win):
    """frame 000000 (454 / 272 sets, two thirds of the slots masked duplicates) on fp32 rows: the split-precision attention core against
    the oracle's gather -> dense_ref.mha core -> scatter, and against the exact-fp32-MFMA kernel of the same plugin, in two logit regimes:
    (a) the stress data of the fp16 test (logit sigma 4.4, tails beyond 20; the softmax turns an absolute logit error into a relative
    probability error) and (b) logits of the size the network produces (Q through the 1 / sqrt(24)-scaled projection, sigma ~1).  Measured
    1.1e-6 / 3.4e-7 of scale at the maximum over a million outputs, 4e-8 / 2e-8 on average -- the fp16 kernel's bound on (a) is 1.5e-3 / 1.5e-4.
    (These bounds found a real defect: hipcc's default -ffp-contract=fast fused "e * inv -> half" into v_fma_mixlo_f16 for the residual but
    converted the fp32-rounded product for the hi fragment; in the rare double-rounding cases hi and lo disagreed by one fp16 ulp: 15 of
    44,000 (row, head) pairs off by 1e-4.  See the comment at the P split in csrc/attention.hip and tools/dbg_attn_split*.py.)"""
    from tests.test_f16_kernels_gpu import _attention_reference
    P, O = pkg.plugin, oracle
    c = cases.caps("ref")
    pts, n = cases.load_frame("000000", c["N"])
    vox = O.points2features(pts, n, cases.p2f_cfg(c))
    rw = O.window_partition(vox["coords"], vox["P"], cases.wp_cfg(c, win))
    gs = O.get_set(rw["gidx"], rw["cinw"], rw["vcnt"], rw["W"], cases.gs_cfg(c, win))
    rng = np.random.default_rng(100 + 2 * win + axis)
    Pn = vox["P"]
    base = rng.standard_normal((Pn, 576)).astype(np.float32)
    for qs, bmax, bmean in ((0.6, 4e-6, 1e-7), (0.6 / np.sqrt(24.0), 1.5e-6, 6e-8)):
        qkv = np.zeros((c["P"], 576), np.float32)
        qkv[:Pn] = base * np.array([qs] * 192 + [1.5] * 192 + [1.0] * 192, np.float32)
        ref = _attention_reference(O, qkv, gs, axis, c["P"])
        got = _run_attention_split(P, c, qkv, gs, axis)
        err = np.abs(got[:Pn] - ref[:Pn])
        scale = max(1.0, np.abs(ref).max())
        print(f"split attention win {win} axis {axis} q-scale {qs:.3f}: max err {err.max():.2e}, mean {err.mean():.2e}, scale {scale:.2f}")
        assert err.max() < bmax * scale and err.mean() < bmean * scale, (qs, err.max(), err.mean())
        assert not got[Pn:].any()
        exact = P.add_set_attention_op(c["W"], 36, 192, 8, axis, c["P"])(dev(qkv[None]), dev(gs["inds"][None]), dev(gs["mask"][None]), scalar(gs["S"]))[0]
        assert np.abs(got - exact[0].cpu().numpy()).max() < bmax * scale


@pytest.mark.parametrize("case", ["empty", "one_voxel", "five_voxels", "full_window", "37_voxels"])
def test_set_attention_split_small_sets(pkg, oracle, case):
    """S = 0; S = 1 with 35 / 31 duplicate slots; a full 12x12 window (4 sets, no duplicates); 37 voxels (2 sets, 35 duplicates)"""
    from tests.test_f16_kernels_gpu import _attention_reference, _synthetic_sets
    P, O = pkg.plugin, oracle
    c = cases.caps("ref")
    cells = {"empty": [], "one_voxel": [(100, 200)], "five_voxels": [(24, 24), (24, 25), (25, 24), (30, 35), (35, 30)],
             "full_window": [(120 + y, 240 + x) for y in range(12) for x in range(12)],
             "37_voxels": [(120 + i // 12, 240 + i % 12) for i in range(37)]}[case]
    Pn, gs = _synthetic_sets(O, c, cells)
    rng = np.random.default_rng(len(cells))
    qkv = np.zeros((c["P"], 576), np.float32); qkv[:Pn] = rng.standard_normal((Pn, 576))
    for axis in (0, 1):
        ref = _attention_reference(O, qkv, gs, axis, c["P"])
        got = _run_attention_split(P, c, qkv, gs, axis)
        assert np.abs(got - ref).max() < 4e-6 * max(1.0, np.abs(ref).max())
        if case == "one_voxel":      # softmax over one live key = that voxel's V row: hi + lo reproduces the fp32 value to 2^-22
            assert np.abs(got[0] - qkv[0, 384:]).max() < 1e-6
