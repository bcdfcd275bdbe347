"""MFMA linear / attention kernels against the fp32 oracle (torch CPU restatement of the
reference's TensorRT layers).  fp32 MFMA is an exact-product fp32 FMA chain; the only
differences from the oracle are summation order => tolerances of a few 1e-6 relative."""
import numpy as np
import pytest
import torch

from tests import cases
from tests.test_plugins_gpu import dev, host, scalar, _sets_for

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.mark.parametrize("K,N,act", [(192, 192, 0), (192, 384, 2), (384, 192, 0), (10, 96, 1), (2, 192, 1), (192, 576, 0)])
def test_linear_plain(pkg, K, N, act):
    P = pkg.plugin
    rng = np.random.default_rng(K * 1000 + N)
    MR, n = 4096, 3001                      # ragged: not a multiple of the 64-row tile
    A = np.zeros((MR, K), np.float32); A[:n] = rng.standard_normal((n, K))
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = (rng.standard_normal(N) * 0.1).astype(np.float32)
    op = P.add_linear_op(W, b, MR, activation=act)
    o = host(op(dev(A[None]), scalar(n))[0])[0]
    y = A[:n].astype(np.float64) @ W.T.astype(np.float64) + b
    if act == 1:
        y = np.maximum(y, 0)
    elif act == 2:
        y = (0.5 + 0.5 * np.tanh(y * (0.035677408136300125 * y * y + 0.7978845608028654))) * y
    assert rel_err(o[:n], y) < 5e-6
    assert not o[n:].any()                  # rows >= count stay zero (reference plugins zero-fill)


def test_linear_fused_epilogues(pkg, oracle):
    """add_cols prologue (q=k=x+pos, v=x) and the chained residual+LayerNorm epilogue."""
    P, O = pkg.plugin, oracle
    rng = np.random.default_rng(77)
    MR, n, C = 8192, 5504, 192
    x = np.zeros((MR, C), np.float32); x[:n] = rng.standard_normal((n, C))
    pos = np.zeros((MR, C), np.float32); pos[:n] = rng.standard_normal((n, C))
    W = (rng.standard_normal((3 * C, C)) / np.sqrt(C)).astype(np.float32); b = (rng.standard_normal(3 * C) * 0.1).astype(np.float32)
    o = host(P.add_linear_op(W, b, MR, add_cols=2 * C)(dev(x[None]), scalar(n), dev(pos[None]))[0])[0]
    xp = (x + pos)[:n].astype(np.float64)
    y = np.concatenate([xp @ W[:2 * C].T.astype(np.float64), x[:n].astype(np.float64) @ W[2 * C:].T.astype(np.float64)], 1) + b
    assert rel_err(o[:n], y) < 5e-6
    # three chained LayerNorm stages, eps = 0 like the reference
    W2 = (rng.standard_normal((C, 2 * C)) / np.sqrt(2 * C)).astype(np.float32); b2 = (rng.standard_normal(C) * 0.1).astype(np.float32)
    h = np.zeros((MR, 2 * C), np.float32); h[:n] = rng.standard_normal((n, 2 * C))
    res = [np.zeros((MR, C), np.float32) for _ in range(3)]
    for r_ in res:
        r_[:n] = rng.standard_normal((n, C))
    lns = [(rng.uniform(0.8, 1.2, C).astype(np.float32), (rng.standard_normal(C) * 0.05).astype(np.float32)) for _ in range(3)]
    op = P.add_linear_op(W2, b2, MR, layer_norms=lns, ln_eps=0.0)
    o = host(op(dev(h[None]), scalar(n), *[dev(r_[None]) for r_ in res])[0])[0]
    y = np.zeros((MR, C), np.float32)
    y[:n] = (h[:n].astype(np.float64) @ W2.T.astype(np.float64) + b2).astype(np.float32)
    for (g, be), r_ in zip(lns, res):
        y = O.layer_norm(y + r_, n, g, be, 0.0)
    assert np.abs(o[:n] - y[:n]).max() < 2e-5
    # serialise / deserialise round trip gives the same bits
    op2 = P.Plugin.deserialize("DsvtLinearPlugin", op.serialize())
    o2 = host(op2(dev(h[None]), scalar(n), *[dev(r_[None]) for r_ in res])[0])[0]
    assert np.array_equal(o, o2)


def _mha_inputs(O, rng, c, vox, rg, axis):
    feat = np.zeros((c["P"], 192), np.float32); feat[:vox["P"]] = rng.standard_normal((vox["P"], 192))
    pos = np.zeros((c["P"], 192), np.float32); pos[:vox["P"]] = rng.standard_normal((vox["P"], 192)) * 0.5
    return feat, pos, O.get_value_by_index(feat, pos, rg["inds"], rg["S"], axis)


@pytest.mark.parametrize("axis", [0, 1])
def test_multi_head_attention_plugin(pkg, oracle, axis):
    """Drop-in MHA op vs the restatement of multHeadAttention() (oracle/dense_ref.py:mha)."""
    from oracle import dense_ref as D
    P, O = pkg.plugin, oracle
    c, vox, rg = _sets_for(O)
    rng = np.random.default_rng(21 + axis)
    w = pkg.synth.make_weights(with_bev=False)
    pre = "module.backbone_3d.stage_0.0.encoder_list.%d.win_attn.self_attn" % axis
    feat, pos, (q, k, v) = _mha_inputs(O, rng, c, vox, rg, axis)
    S = rg["S"]
    ref = D.mha(D.T(q[:S]), D.T(k[:S]), D.T(v[:S]), D.T(rg["mask0_h"][:S]), w, pre).numpy()
    op = P.add_multi_head_attention_op(w[pre + ".in_proj_weight"], w[pre + ".in_proj_bias"], w[pre + ".out_proj.weight"],
                                       w[pre + ".out_proj.bias"], c["W"], 36, 192, 8)
    o = host(op(dev(q[None]), dev(k[None]), dev(v[None]), dev(rg["mask0_h"][None]), scalar(S))[0])[0]
    assert np.abs(o[:S] - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())
    assert not o[S:].any()


@pytest.mark.parametrize("axis", [0, 1])
def test_fused_set_attention_equals_unfused_chain(pkg, oracle, axis):
    """Linear(add_cols) on voxel rows + DsvtSetAttention + out-proj == GetValueByIndex -> MHA ->
    MapSetFeature2Voxel of the reference wiring (src/dsvt-ai-trt.cpp:653-663)."""
    from oracle import dense_ref as D
    P, O = pkg.plugin, oracle
    c, vox, rg = _sets_for(O)
    rng = np.random.default_rng(31 + axis)
    w = pkg.synth.make_weights(with_bev=False)
    pre = "module.backbone_3d.stage_0.1.encoder_list.%d.win_attn.self_attn" % axis
    feat, pos, (q, k, v) = _mha_inputs(O, rng, c, vox, rg, axis)
    S, Pn, C = rg["S"], vox["P"], 192
    a = D.mha(D.T(q[:S]), D.T(k[:S]), D.T(v[:S]), D.T(rg["mask0_h"][:S]), w, pre).numpy()
    a_full = np.zeros((c["W"], 36, C), np.float32); a_full[:S] = a
    ref = O.map_set_feature2voxel(a_full, rg["inds"], S, axis, c["P"])
    wi, bi = w[pre + ".in_proj_weight"].copy(), w[pre + ".in_proj_bias"].copy()
    wi[:C] /= np.float32(np.sqrt(24.0)); bi[:C] /= np.float32(np.sqrt(24.0))       # q / sqrt(head_dim), :386-405
    qkv = P.add_linear_op(wi, bi, c["P"], add_cols=2 * C)(dev(feat[None]), scalar(Pn), dev(pos[None]))[0]
    att = P.add_set_attention_op(c["W"], 36, C, 8, axis, c["P"])(qkv, dev(rg["inds"][None]), dev(rg["mask"][None]), scalar(S))[0]
    out = P.add_linear_op(w[pre + ".out_proj.weight"], w[pre + ".out_proj.bias"], c["P"])(att, scalar(Pn))[0]
    o = host(out)[0]
    assert np.abs(o[:Pn] - ref[:Pn]).max() < 2e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("K,N,act,nln", [(192, 192, 0, 1), (192, 384, 2, 0), (384, 192, 0, 3), (192, 576, 0, 0)])
def test_linear_f16_mfma(pkg, oracle, K, N, act, nln):
    """fp16-operand MFMA variant (compute_type=1): inputs rounded to fp16 (2^-11 relative), fp32
    accumulation and epilogues => 2e-3 of the output scale; must equal the fp32 kernel run on
    pre-rounded operands to fp32 summation-order noise."""
    P = pkg.plugin
    rng = np.random.default_rng(K + N + act)
    MR, n = 8192, 5504
    A = np.zeros((MR, K), np.float32); A[:n] = rng.standard_normal((n, K))
    A2 = np.zeros((MR, K), np.float32); A2[:n] = rng.standard_normal((n, K)) * 0.5
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32); b = (rng.standard_normal(N) * 0.1).astype(np.float32)
    res = [np.zeros((MR, N), np.float32) for _ in range(nln)]
    for r_ in res:
        r_[:n] = rng.standard_normal((n, N))
    lns = [(rng.uniform(0.8, 1.2, N).astype(np.float32), (rng.standard_normal(N) * 0.05).astype(np.float32)) for _ in range(nln)]
    add_cols = 384 if N == 576 else 0
    extra = ([dev(A2[None])] if add_cols else []) + [dev(r_[None]) for r_ in res]
    kw = dict(activation=act, add_cols=add_cols, layer_norms=lns)
    o16 = host(P.add_linear_op(W, b, MR, compute_type=P.COMPUTE_F16, **kw)(dev(A[None]), scalar(n), *extra)[0])[0]
    o32 = host(P.add_linear_op(W, b, MR, compute_type=P.COMPUTE_F32, **kw)(dev(A[None]), scalar(n), *extra)[0])[0]
    scale = np.abs(o32[:n]).max()
    assert np.abs(o16[:n] - o32[:n]).max() < 2e-3 * scale
    assert not o16[n:].any()
    if add_cols == 0:
        # same kernel maths on operands that are already fp16-representable: only summation order differs
        Ah = A.astype(np.float16).astype(np.float32); Wh = W.astype(np.float16).astype(np.float32)
        e16 = host(P.add_linear_op(Wh, b, MR, compute_type=P.COMPUTE_F16, **kw)(dev(Ah[None]), scalar(n), *extra)[0])[0]
        e32 = host(P.add_linear_op(Wh, b, MR, compute_type=P.COMPUTE_F32, **kw)(dev(Ah[None]), scalar(n), *extra)[0])[0]
        assert np.abs(e16[:n] - e32[:n]).max() < 2e-5 * scale


@pytest.mark.parametrize("block_ln,MR,n", [(False, 8192, 5504), (True, 8192, 5504),
                                             (True, 65536, 34483),      # nine live waves per workgroup (one workgroup per CU)
                                             (False, 65536, 39000),     # ten
                                             (False, 65536, 50000)])    # beyond 160 rows per CU: eight-wave workgroups in rounds
def test_fused_encoder_mlp_equals_linear_chain(pkg, block_ln, MR, n):
    """DsvtEncoderMlpPlugin (one launch, activations chained through registers with a permuted-k operand
    layout) against the same maths as three DsvtLinear launches in fp16 mode."""
    P = pkg.plugin
    rng = np.random.default_rng(11 + block_ln)
    C = 192
    w = pkg.synth.make_weights(with_bev=False)
    lp = "module.backbone_3d.stage_0.2.encoder_list.1"
    ln = lambda k: (w[lp + k + ".weight"], w[lp + k + ".bias"])
    lns = [ln(".win_attn.norm1"), ln(".win_attn.norm2"), ln(".norm")]
    if block_ln:
        lns.append((w["module.backbone_3d.residual_norm_stage_0.2.weight"], w["module.backbone_3d.residual_norm_stage_0.2.bias"]))
    att = np.zeros((MR, C), np.float32); att[:n] = rng.standard_normal((n, C))
    x = np.zeros((MR, C), np.float32); x[:n] = rng.standard_normal((n, C))
    xb = np.zeros((MR, C), np.float32); xb[:n] = rng.standard_normal((n, C))
    att_h = dev(att[None]).half(); cnt = scalar(n)
    f16 = dict(compute_type=P.COMPUTE_F16, input_half=True)
    out = P.add_linear_op(w[lp + ".win_attn.self_attn.out_proj.weight"], w[lp + ".win_attn.self_attn.out_proj.bias"], MR,
                          layer_norms=lns[:1], output_mode=P.OUT_BOTH, **f16)
    fc1 = P.add_linear_op(w[lp + ".win_attn.linear1.weight"], w[lp + ".win_attn.linear1.bias"], MR, activation=P.ACT_GELU,
                          output_mode=P.OUT_F16, **f16)
    fc2 = P.add_linear_op(w[lp + ".win_attn.linear2.weight"], w[lp + ".win_attn.linear2.bias"], MR, layer_norms=lns[1:],
                          output_mode=P.OUT_BOTH, **f16)
    s1, s1h = out(att_h, cnt, dev(x[None]))
    h = fc1(s1h, cnt)[0]
    res = [s1, dev(x[None])] + ([dev(xb[None])] if block_ln else [])
    ref, ref_h = fc2(h, cnt, *res)
    mlp = P.add_encoder_mlp_op(w[lp + ".win_attn.self_attn.out_proj.weight"], w[lp + ".win_attn.self_attn.out_proj.bias"],
                               w[lp + ".win_attn.linear1.weight"], w[lp + ".win_attn.linear1.bias"],
                               w[lp + ".win_attn.linear2.weight"], w[lp + ".win_attn.linear2.bias"], lns, MR)
    args = [att_h, cnt, dev(x[None])] + ([dev(xb[None])] if block_ln else [])
    got, got_h = mlp(*args)
    torch.cuda.synchronize()
    g, r_ = host(got)[0], host(ref)[0]
    # same fp16 roundings of the operands; differences: summation order and the chain keeps s1 (LN2's residual) in fp32
    assert np.abs(g[:n] - r_[:n]).max() < 3e-3
    assert np.abs(g[:n] - r_[:n]).mean() < 2e-4
    assert not g[n:].any()
    assert np.abs(got_h[0, :n].float().cpu().numpy() - g[:n]).max() < 4e-3       # fp16 copy of the same result


def test_linear_posembed_prologue_equals_two_launches(pkg):
    """pe_weight / pe_bias: the K_in = 2 FC + BN + ReLU of the position-embedding MLP (src/dsvt-ai-trt.cpp:461-492)
    evaluated in the operand prologue of the second FC == the two-launch chain with an fp16 intermediate."""
    P = pkg.plugin
    rng = np.random.default_rng(5)
    MR, n, C = 8192, 5504, 192
    xy = np.zeros((MR, 2), np.float32); xy[:n] = rng.uniform(-6, 6, (n, 2))
    Wa = (rng.standard_normal((C, 2)) * 0.5).astype(np.float32); ba = (rng.standard_normal(C) * 0.2).astype(np.float32)
    Wb = (rng.standard_normal((C, C)) / np.sqrt(C)).astype(np.float32); bb = (rng.standard_normal(C) * 0.1).astype(np.float32)
    cnt = scalar(n)
    h = P.add_linear_op(Wa, ba, MR, activation=P.ACT_RELU)(dev(xy[None]), cnt)[0]       # fp32 kernel (K = 2)
    ref = host(P.add_linear_op(Wb, bb, MR, compute_type=P.COMPUTE_F16)(h, cnt)[0])[0]
    fused = P.add_linear_op(Wb, bb, MR, compute_type=P.COMPUTE_F16, pe_weight=Wa, pe_bias=ba)
    got = host(fused(dev(xy[None]), cnt)[0])[0]
    scale = np.abs(ref[:n]).max()
    # the hidden row is rounded to fp16 in both; fma contraction of the K = 2 dot product may move it by one fp16 ulp
    assert np.abs(got[:n] - ref[:n]).max() < 2e-3 * scale
    assert np.abs(got[:n] - ref[:n]).mean() < 1e-4 * scale
    assert not got[n:].any()
    # serialise -> deserialise -> same bytes and same result
    blob = fused.serialize()
    again = P.Plugin.deserialize("DsvtLinearPlugin", blob)
    assert again.serialize() == blob
    assert np.array_equal(host(again(dev(xy[None]), cnt)[0])[0], got)
    # the prologue exists only in the fp16 kernel: the fp32 compute type refuses the fields
    with pytest.raises(Exception):
        P.add_linear_op(Wb, bb, MR, compute_type=P.COMPUTE_F32, pe_weight=Wa, pe_bias=ba)


def test_batched_pos_embed_equals_per_layer_launches(pkg):
    """DsvtPosEmbedPlugin (all layers, one launch, blockIdx.z = layer) == one DsvtLinearPlugin launch per layer with the
    position-embedding prologue: same kernel body, so bit-identical."""
    P = pkg.plugin
    rng = np.random.default_rng(9)
    MR, n, C, L = 8192, 5504, 192, 4
    xys = [np.zeros((MR, 2), np.float32) for _ in range(2)]
    for x in xys:
        x[:n] = rng.uniform(-12, 12, (n, 2))
    Wa = [(rng.standard_normal((C, 2)) * 0.5).astype(np.float32) for _ in range(L)]
    ba = [(rng.standard_normal(C) * 0.2).astype(np.float32) for _ in range(L)]
    Wb = [(rng.standard_normal((C, C)) / np.sqrt(C)).astype(np.float32) for _ in range(L)]
    bb = [(rng.standard_normal(C) * 0.1).astype(np.float32) for _ in range(L)]
    src = [0, 1, 0, 1]
    cnt = scalar(n)
    dx = [dev(x[None]) for x in xys]
    outs = P.add_pos_embed_op(MR, src, Wa, ba, Wb, bb)(cnt, *dx)
    torch.cuda.synchronize()
    assert len(outs) == L
    for l in range(L):
        ref = P.add_linear_op(Wb[l], bb[l], MR, compute_type=P.COMPUTE_F16, output_mode=P.OUT_F16, pe_weight=Wa[l], pe_bias=ba[l])(dx[src[l]], cnt)[0]
        assert torch.equal(outs[l], ref)
        assert not outs[l][0, n:].any()


def test_qkv_position_table_gather(pkg):
    """QKV linear with add_gather_width: q = k = (x + table[y * wx + x_cell]) W^T, bit-identical to the same op fed the gathered
    position rows as a full [rows, K] tensor (the table is what the position-embedding MLP yields on the window's cell grid)."""
    DEV = "cuda:0"
    P = pkg.plugin
    rng = np.random.default_rng(5)
    MR, n, C, wx, wy = 4096, 3000, 192, 24, 24
    x = torch.from_numpy(rng.standard_normal((1, MR, C)).astype(np.float32)).half().to(DEV)
    table = torch.from_numpy(rng.standard_normal((1, wx * wy, C)).astype(np.float32)).half().to(DEV)
    c2d = np.zeros((1, MR, 3), np.int32)
    c2d[0, :, 1] = rng.integers(0, wy, MR); c2d[0, :, 2] = rng.integers(0, wx, MR)
    cell = c2d[0, :, 1] * wx + c2d[0, :, 2]
    pos = table[0][torch.from_numpy(cell).to(DEV).long()][None].contiguous()
    W = (rng.standard_normal((3 * C, C)) / np.sqrt(C)).astype(np.float32); b = (rng.standard_normal(3 * C) * 0.1).astype(np.float32)
    kw = dict(add_cols=2 * C, compute_type=P.COMPUTE_F16, input_half=True, output_mode=P.OUT_F16)
    cnt = torch.tensor([n], dtype=torch.int32, device=DEV)
    ref = P.add_linear_op(W, b, MR, **kw)(x, cnt, pos)[0]
    op = P.add_linear_op(W, b, MR, add_gather_width=wx, **kw)
    got = op(x, cnt, table, torch.from_numpy(c2d).to(DEV))[0]
    torch.cuda.synchronize()
    assert torch.equal(got[0, :n], ref[0, :n])
    again = P.Plugin.deserialize("DsvtLinearPlugin", op.serialize())(x, cnt, table, torch.from_numpy(c2d).to(DEV))[0]
    torch.cuda.synchronize()
    assert torch.equal(again[0, :n], ref[0, :n])


@pytest.mark.parametrize("MR,n", [(65536, 34483), (65536, 39000), (65536, 50000), (8192, 100)])
def test_qkv_rows_kernel_row_regimes(pkg, MR, n):
    """The whole-row QKV kernel picks 8, 9 or 10 live waves of 16 rows from the device-side count (one workgroup per CU), and runs
    eight-wave workgroups in rounds beyond 160 rows per CU: every regime against an fp32 torch product of the same fp16 operands."""
    P = pkg.plugin
    DEV = "cuda:0"
    g = torch.Generator(device="cpu").manual_seed(MR + n)
    C = 192
    x = torch.randn((1, MR, C), generator=g).half().to(DEV)
    pos = (torch.randn((1, MR, C), generator=g) * 0.5).half().to(DEV)
    W = (torch.randn((3 * C, C), generator=g) / np.sqrt(C)); b = torch.randn(3 * C, generator=g) * 0.1
    op = P.add_linear_op(W.numpy(), b.numpy(), MR, add_cols=2 * C, compute_type=P.COMPUTE_F16, input_half=True, output_mode=P.OUT_F16)
    got = op(x, torch.tensor([n], dtype=torch.int32, device=DEV), pos)[0]
    torch.cuda.synchronize()
    Wd = W.half().float().to(DEV); xs = (x[0, :n] + pos[0, :n]).float(); xv = x[0, :n].float()
    ref = torch.cat([xs @ Wd[:2 * C].T, xv @ Wd[2 * C:].T], 1) + b.to(DEV)
    err = (got[0, :n].float() - ref).abs().max().item()
    assert err < 2e-2 * ref.abs().max().item()
    assert not got[0, n:].any()
