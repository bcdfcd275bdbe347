"""fp32 PyTorch-CPU restatement of the dense (TensorRT-API) layers of the reference
network and of the full frame pipeline, wired exactly as `createEngine`
(src/dsvt-ai-trt.cpp:532-1762) wires them.  The plugin ops are the C oracle
(oracle/dsvt_oracle.c) called through oracle/oracle.py.

TEST INFRASTRUCTURE ONLY (see dsvt_oracle.c header).

PARITY UNPINNED for the layers in this file: in the reference they are
TensorRT 8.2.1.8 library layers (addFullyConnected / addMatrixMultiply /
addSoftMax / addScale / addConvolutionNd / addDeconvolutionNd / addTopK /
addGather / addUnary; call sites listed in SURVEY.md section 8c).  TensorRT is a
binary dependency that is not in the tree, the reference has no tests or golden
vectors at that boundary and its weight file `dsvt.wts` is missing, so nothing
pins these restatements except the published layer semantics
(y = xW^T + b, BN folded to scale/shift, softmax over the last axis, ...).
Top-K tie order is TensorRT's and is not reproducible: callers use tie-free
synthetic heat maps.
"""
import math
import numpy as np
import torch
import torch.nn.functional as F

from . import oracle as O


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32))


# ---- src/dsvt-ai-trt.cpp:99-122, 149-180: BN folded to (scale, shift) ----------------
def bn_fold(w, prefix, eps):
    g, b = w[prefix + ".weight"], w[prefix + ".bias"]
    m, v = w[prefix + ".running_mean"], w[prefix + ".running_var"]
    eps = np.float32(eps)
    scale = (g / np.sqrt(v + eps)).astype(np.float32)
    shift = (b - m * g / np.sqrt(v + eps)).astype(np.float32)
    return scale, shift


def fc_bn_relu(x, w, lin, bn, eps=1e-5, bias=False):
    """fullyConnectedBnLELU :268-286 (PFN: no bias)."""
    y = F.linear(x, T(w[lin + ".weight"]), T(w[lin + ".bias"]) if bias else None)
    s, sh = bn_fold(w, bn, eps)
    return torch.relu(y * T(s) + T(sh))


def posembed(xy, w, prefix):
    """fullyConnectedBnLELU_fullyConnected :461-492: FC 2->192 (+bias), BN1d(1e-5), ReLU, FC 192->192 (+bias)."""
    h = fc_bn_relu(xy, w, prefix + ".0", prefix + ".1", 1e-5, bias=True)
    return F.linear(h, T(w[prefix + ".3.weight"]), T(w[prefix + ".3.bias"]))


def mha(q, k, v, mask_h, w, prefix, num_heads=8):
    """multHeadAttention :288-458.  q,k,v [S,L,C]; mask_h [S,H,L]  ->  [S,L,C]."""
    S, L, Cn = q.shape
    dh = Cn // num_heads
    wi, bi = w[prefix + ".in_proj_weight"], w[prefix + ".in_proj_bias"]
    Q = F.linear(q, T(wi[:Cn]), T(bi[:Cn]))            # helper.h:369-433 row split
    K = F.linear(k, T(wi[Cn:2 * Cn]), T(bi[Cn:2 * Cn]))
    V = F.linear(v, T(wi[2 * Cn:]), T(bi[2 * Cn:]))
    Q = Q.view(S, L, num_heads, dh).permute(0, 2, 1, 3)        # [S,H,L,dh]  :352-372
    K = K.view(S, L, num_heads, dh).permute(0, 2, 1, 3)
    V = V.view(S, L, num_heads, dh).permute(0, 2, 1, 3)
    Q = Q / torch.tensor(math.sqrt(Cn / num_heads), dtype=torch.float32)   # :386-405 elementwise DIV by sqrt(24)
    att = Q @ K.transpose(-1, -2)                              # :410
    att = att + mask_h[:, :, None, :]                          # :376-382, :412 broadcast over queries
    att = torch.softmax(att, dim=-1)                           # :414-415
    o = att @ V                                                # :417
    o = o.permute(0, 2, 1, 3).reshape(S, L, Cn)
    return F.linear(o, T(w[prefix + ".out_proj.weight"]), T(w[prefix + ".out_proj.bias"]))   # :448


def np_(t):
    return t.detach().numpy()


class OracleCfg:
    """Runtime caps (the reference's params.h macros) for one pipeline instance."""

    def __init__(self, max_points=50000, max_points_filter=30000, max_pillars=10000, max_win=800,
                 max_vox_per_win=576, top_k=500, score_threshold=0.3, ln_eps=0.0, blocks=4, max_sets=None):
        self.p2f = dict(max_points_num=max_points, max_points_num_voxel_filter=max_points_filter,
                        max_pillars_num=max_pillars, point_feature_num=4, feature_num=10,
                        max_num_points_per_voxel=48,
                        point_cloud_range=[-74.88, -74.88, -5.0, 74.88, 74.88, 3.0],
                        voxel_size=[0.32, 0.32, 8.0], grid_size=[468, 468, 1])
        self.wp = [dict(max_win_num=max_win, max_voxel_num_per_win=max_vox_per_win,
                        sparse_shape=[468, 468, 1], win_shape=[12, 12, 1], shift_list=[0, 0, 0],
                        max_pillars_num=max_pillars),
                   dict(max_win_num=max_win, max_voxel_num_per_win=max_vox_per_win,
                        sparse_shape=[468, 468, 1], win_shape=[24, 24, 1], shift_list=[6, 6, 0],
                        max_pillars_num=max_pillars)]
        # the reference sizes the set dimension with MAX_WIN_NUM (getSet.cu:147,242); max_sets only enlarges that dimension
        self.gs = [dict(max_win_num=max_sets or max_win, max_voxel_num_per_win=max_vox_per_win, voxel_num_set=36,
                        win_shape=c["win_shape"]) for c in self.wp]
        self.fb = dict(max_top_k=top_k, point_cloud_range=[-74.88, 74.88, -74.88, 74.88, -5.0, 3.0],
                       voxel_size=[0.32, 0.32, 8.0], score_threshold=score_threshold)
        self.max_pillars, self.max_points_filter, self.ln_eps, self.blocks = max_pillars, max_points_filter, ln_eps, blocks


def voxel_stage(points, n, w, cfg):
    """Points2Features -> PFN -> ScatterMax x2 -> WindowPartition x2 -> GetSet x2 -> pos-embeds.
    src/dsvt-ai-trt.cpp:571-637."""
    v = O.points2features(points, n, cfg.p2f)
    P, Nk = v["P"], v["Nk"]
    MP, MN = cfg.max_pillars, cfg.max_points_filter
    f0 = T(v["feat"][:Nk])
    x0 = fc_bn_relu(f0, w, "module.vfe.pfn_layers.0.linear", "module.vfe.pfn_layers.0.norm")       # :577
    x0p = np.zeros((MN, 96), np.float32); x0p[:Nk] = np_(x0)
    mp0, _ = O.scatter_max(x0p, v["pidx"], v["pcnt"], P, MN, MP, 96)                                 # :579
    cat = torch.cat([x0, T(mp0[:Nk])], 1)                                                           # :583-585
    x1 = fc_bn_relu(cat, w, "module.vfe.pfn_layers.1.linear", "module.vfe.pfn_layers.1.norm")      # :587
    x1p = np.zeros((MN, 192), np.float32); x1p[:Nk] = np_(x1)
    _, vfeat = O.scatter_max(x1p, v["pidx"], v["pcnt"], P, MN, MP, 192)                              # :589
    wps = [O.window_partition(v["coords"], P, c) for c in cfg.wp]                                   # :592-597
    gss = [O.get_set(wp["gidx"], wp["cinw"], wp["vcnt"], wp["W"], c) for wp, c in zip(wps, cfg.gs)]  # :598-601
    pe = {}
    for b in range(cfg.blocks):
        for l in range(2):                                                                          # :603-637
            pre = f"module.backbone_3d.input_layer.posembed_layers.0.{b}.{l}.position_embedding_head"
            full = np.zeros((MP, 192), np.float32)
            full[:P] = np_(posembed(T(wps[l]["xy"][:P]), w, pre))
            pe[(b, l)] = full
    return dict(vox=v, vfeat=vfeat, wps=wps, gss=gss, pe=pe, P=P, Nk=Nk)


def dsvt_layer(x, x_pos, gs, axis, P, w, prefix, cfg):
    """One encoder layer, src/dsvt-ai-trt.cpp:653-697.  x [MP,192] numpy (rows >= P zero)."""
    S = gs["S"]
    q, k, v = O.get_value_by_index(x, x_pos, gs["inds"], S, axis)                                   # :653
    a = mha(T(q[:S]), T(k[:S]), T(v[:S]), T(gs["mask0_h"][:S]), w, prefix + ".win_attn.self_attn")  # :657 (always output 3)
    a_full = np.zeros_like(q); a_full[:S] = np_(a)
    y = O.map_set_feature2voxel(a_full, gs["inds"], S, axis, x.shape[0])                            # :663
    s1 = O.layer_norm(y + x, P, w[prefix + ".win_attn.norm1.weight"], w[prefix + ".win_attn.norm1.bias"], cfg.ln_eps)  # :669-676
    h = F.linear(T(s1[:P]), T(w[prefix + ".win_attn.linear1.weight"]), T(w[prefix + ".win_attn.linear1.bias"]))       # :506
    hp = np.zeros((x.shape[0], h.shape[1]), np.float32); hp[:P] = np_(h)
    g = O.gelu(hp, P)
    f2 = F.linear(T(g[:P]), T(w[prefix + ".win_attn.linear2.weight"]), T(w[prefix + ".win_attn.linear2.bias"]))      # :525
    f2p = np.zeros_like(x); f2p[:P] = np_(f2)
    s2 = O.layer_norm(s1 + f2p, P, w[prefix + ".win_attn.norm2.weight"], w[prefix + ".win_attn.norm2.bias"], cfg.ln_eps)  # :684-690
    return O.layer_norm(s2 + x, P, w[prefix + ".norm.weight"], w[prefix + ".norm.bias"], cfg.ln_eps)                 # :691-697


def dsvt_blocks(st, w, cfg, nblocks=None, trace=None, stage=0):
    """stage: which stage's weights (`stage_{stage}` / `residual_norm_stage_{stage}`); the reference has stage 0 only (src/dsvt-ai-trt.cpp:653-756)"""
    x = st["vfeat"].copy()
    P = st["P"]
    x[P:] = 0
    for b in range(cfg.blocks if nblocks is None else nblocks):
        xb = x
        gs = st["gss"][b % 2]
        for l in range(2):
            x = dsvt_layer(x, st["pe"][(b, l)], gs, l, P, w, f"module.backbone_3d.stage_{stage}.{b}.encoder_list.{l}", cfg)
            if trace is not None:
                trace[(b, l)] = x.copy()
        x = O.layer_norm(x + xb, P, w[f"module.backbone_3d.residual_norm_stage_{stage}.{b}.weight"],
                         w[f"module.backbone_3d.residual_norm_stage_{stage}.{b}.bias"], cfg.ln_eps)       # :750-756
        if trace is not None:
            trace[(b, "res")] = x.copy()
    return x


def conv_bn(x, w, conv, bn, stride, pad, relu):
    y = F.conv2d(x, T(w[conv + ".weight"]), None, stride, pad)
    s, sh = bn_fold(w, bn, 1e-3)                                                                    # :191,208
    y = y * T(s)[None, :, None, None] + T(sh)[None, :, None, None]
    return torch.relu(y) if relu else y


def bev_backbone(bev_nchw, w):
    """src/dsvt-ai-trt.cpp:1144-1364."""
    blocks = [(0, 1, 2), (1, 2, 3), (2, 2, 3)]
    deb = [(0, 1), (1, 2), (2, 4)]
    x = bev_nchw
    ups = []
    for (i, stride, nb) in blocks:
        for j in range(nb):
            p = f"module.backbone_2d.blocks.{i}.{j}"
            s = stride if j == 0 else 1
            y = conv_bn(x, w, p + ".conv1", p + ".bn1", s, 1, True)
            y = conv_bn(y, w, p + ".conv2", p + ".bn2", 1, 1, False)
            idn = conv_bn(x, w, p + ".downsample_layer.0", p + ".downsample_layer.1", s, 0, False) if j == 0 else x
            x = torch.relu(y + idn)
        k = deb[i][1]
        p = f"module.backbone_2d.deblocks.{i}"
        u = F.conv_transpose2d(x, T(w[p + ".0.weight"]), None, stride=k)                            # :217-246
        s_, sh_ = bn_fold(w, p + ".1", 1e-3)
        ups.append(torch.relu(u * T(s_)[None, :, None, None] + T(sh_)[None, :, None, None]))
    return torch.cat(ups, 1)                                                                        # :1363


def center_head(x, w):
    """src/dsvt-ai-trt.cpp:1369-1468."""
    sh = conv_bn(x, w, "module.dense_head.shared_conv.0", "module.dense_head.shared_conv.1", 1, 1, True)
    out = {}
    for name in ["center", "center_z", "dim", "rot", "hm"]:          # iou head is built but unused (:1440-1452)
        p = f"module.dense_head.heads_list.0.{name}"
        h = conv_bn(sh, w, p + ".0.0", p + ".0.1", 1, 1, True)
        out[name] = F.conv2d(h, T(w[p + ".1.weight"]), T(w[p + ".1.bias"]), 1, 1)
    return out


def postprocess(heads, top_k=500):
    """src/dsvt-ai-trt.cpp:1479-1669.  Returns the eight FilterBoxByScore inputs."""
    hm = torch.sigmoid(heads["hm"])[0]                   # [10,H,W]   :1479
    ncls, H, Wd = hm.shape
    dim = torch.exp(heads["dim"])[0]                     # :1487
    sc1, idx1 = torch.topk(hm.reshape(ncls, H * Wd), top_k, dim=1)          # :1519
    sc2, idx2 = torch.topk(sc1.reshape(-1), top_k)                           # :1561
    cls = idx2 // top_k                                                      # :1573
    ind = idx1.reshape(-1)[idx2]                                             # :1588
    ys, xs = ind // Wd, ind % Wd                                             # :1540-1547
    g = lambda t: t.reshape(t.shape[0], -1)[:, ind].T.contiguous()           # gather at ind, -> [K, ch]
    center = g(heads["center"][0]); center_z = g(heads["center_z"][0]); dim_g = g(dim)
    rot = g(heads["rot"][0])
    angle = torch.atan(rot[:, 1:2] / rot[:, 0:1])                            # :1668-1669  atan(sin/cos), slices :1494-1501
    return dict(scores=np_(sc2), classes=np_(cls).astype(np.uint32), xs=np_(xs).astype(np.uint32),
                ys=np_(ys).astype(np.uint32), center=np_(center), center_z=np_(center_z),
                angle=np_(angle), dim=np_(dim_g))


def forward(points, n, w, cfg, trace=None):
    """Whole reference network on one frame -> (boxes[top_k,9], count)."""
    st = voxel_stage(points, n, w, cfg)
    x = dsvt_blocks(st, w, cfg, trace=trace)
    bev = O.map2bev(x, st["vox"]["coords"], st["P"], 468, 468)               # :1128
    bev = T(bev).permute(2, 0, 1)[None]                                      # :1131-1133 NHWC -> NCHW
    feat = bev_backbone(bev, w)
    heads = center_head(feat, w)
    pp = postprocess(heads, cfg.fb["max_top_k"])
    boxes, cnt = O.filter_box_by_score(pp["scores"], pp["classes"], pp["xs"], pp["ys"], pp["center"],
                                       pp["center_z"], pp["angle"], pp["dim"], cfg.fb)
    if trace is not None:
        trace["state"], trace["x"], trace["heads"], trace["pp"] = st, x, heads, pp
    return boxes, cnt


# ---------------------------------------------------------------------------------------------------------------------
# stage reduction of a multi-stage 3-D voxel DSVT.  NOT in the reference (its voxel z index is forced to 0,
# plugins/src/points2Features.cu:689-690,755): this restates UPSTREAM DSVT's published semantics (DSVTInputLayer pooling indices +
# Stage_ReductionAtt_Block: MaxPool1d query, nn.MultiheadAttention(query, prepool + pos_embedding, prepool, key_padding_mask), residual,
# LayerNorm) and is PARITY-UNPINNED: nothing the reference holds can confirm it.
# ---------------------------------------------------------------------------------------------------------------------
def pool_partition(coords, P, sparse_shape, stride):
    """coords [.., 4] (b, z, y, x) of the P voxels of a stage (any order) -> (pooled coords [P2, 4] ascending by pooled cell key (b, z, y, x),
    child table [P2, pool_volume] (input row or -1), parent [P]).  Slot of a child: (x % sx) sy sz + (y % sy) sz + z % sz (upstream's voxel_index_in_win)."""
    gx, gy, gz = sparse_shape; sx, sy, sz = stride
    c = np.asarray(coords[:P], np.int64)
    px, py, pz = -(-gx // sx), -(-gy // sy), -(-gz // sz)
    z2, y2, x2 = c[:, 1] // sz, c[:, 2] // sy, c[:, 3] // sx
    key = ((c[:, 0] * pz + z2) * py + y2) * px + x2
    uniq, parent = np.unique(key, return_inverse=True)
    pv = sx * sy * sz
    slot = ((c[:, 3] % sx) * sy + (c[:, 2] % sy)) * sz + (c[:, 1] % sz)
    table = np.full((len(uniq), pv), -1, np.int32)
    table[parent, slot] = np.arange(P, dtype=np.int32)
    coords2 = np.zeros((len(uniq), 4), np.int32)
    coords2[:, 3] = uniq % px; coords2[:, 2] = (uniq // px) % py; coords2[:, 1] = (uniq // (px * py)) % pz; coords2[:, 0] = uniq // (px * py * pz)
    return coords2, table, parent.astype(np.int32)


def stage_reduction_att(x, table, w, prefix, num_heads=8, eps=1e-5):
    """x [P, C] voxel features, table [P2, pv] -> pooled features [P2, C] (Stage_ReductionAtt_Block.forward)"""
    xt = T(np.asarray(x, np.float32))
    tb = torch.from_numpy(np.asarray(table, np.int64))
    P2, pv = tb.shape
    C = xt.shape[1]
    pre = torch.zeros((P2, pv, C), dtype=torch.float32)
    valid = tb >= 0
    pre[valid] = xt[tb[valid]]
    src = pre.max(dim=1).values                                              # MaxPool1d(pool_volume): the zero rows of empty slots take part
    key = pre + T(w[prefix + ".pos_embedding"])[None]
    wi, bi = T(w[prefix + ".self_attn.in_proj_weight"]), T(w[prefix + ".self_attn.in_proj_bias"])
    hd = C // num_heads
    q = (F.linear(src, wi[:C], bi[:C]) / float(np.sqrt(hd))).reshape(P2, num_heads, hd)
    k = F.linear(key, wi[C:2 * C], bi[C:2 * C]).reshape(P2, pv, num_heads, hd)
    v = F.linear(pre, wi[2 * C:], bi[2 * C:]).reshape(P2, pv, num_heads, hd)
    sc = torch.einsum("phd,pjhd->phj", q, k)
    sc = sc.masked_fill(~valid[:, None, :], float("-inf"))
    a = torch.softmax(sc, dim=-1)
    ctx = torch.einsum("phj,pjhd->phd", a, v).reshape(P2, C)
    out = F.linear(ctx, T(w[prefix + ".self_attn.out_proj.weight"]), T(w[prefix + ".self_attn.out_proj.bias"]))
    y = F.layer_norm(src + out, (C,), T(w[prefix + ".norm.weight"]), T(w[prefix + ".norm.bias"]), eps)
    return np_(y)


def backbone_3d(points, n, w, grid, voxel_size, windows, strides, max_points, max_voxels, max_win, max_sets, trace=None):
    """The multi-stage 3-D voxel backbone dsvt-ai-trt_amd/pipeline3d.py builds, on the CPU: z-aware Points2Features + PFN (restated reference kernels),
    per stage WindowPartition / GetSet / one DSVT block (restated reference kernels and layers) with a position-embedding MLP over (x, y, z), and
    between stages pool_partition + stage_reduction_att (upstream semantics, parity-unpinned).  Returns (features [P_last, C], coords [P_last, 4])."""
    MP = max_voxels
    p2f = dict(max_points_num=max_points, max_points_num_voxel_filter=max_points, max_pillars_num=MP, point_feature_num=4, feature_num=10,
               max_num_points_per_voxel=48, point_cloud_range=[-74.88, -74.88, -5.0, 74.88, 74.88, 3.0], voxel_size=list(voxel_size), grid_size=list(grid))
    v = O.points2features(points, n, p2f)
    P, Nk = v["P"], v["Nk"]
    x0 = fc_bn_relu(T(v["feat"][:Nk]), w, "module.vfe.pfn_layers.0.linear", "module.vfe.pfn_layers.0.norm")
    x0p = np.zeros((max_points, 96), np.float32); x0p[:Nk] = np_(x0)
    mp0, _ = O.scatter_max(x0p, v["pidx"], v["pcnt"], P, max_points, MP, 96)
    x1 = fc_bn_relu(torch.cat([x0, T(mp0[:Nk])], 1), w, "module.vfe.pfn_layers.1.linear", "module.vfe.pfn_layers.1.norm")
    x1p = np.zeros((max_points, 192), np.float32); x1p[:Nk] = np_(x1)
    _, x = O.scatter_max(x1p, v["pidx"], v["pcnt"], P, max_points, MP, 192)
    coords = v["coords"].astype(np.int32)
    g = tuple(grid)
    cfg = OracleCfg(max_pillars=MP, blocks=1)
    for s_, win in enumerate(windows):
        wx, wy, wz = win
        wcfg = dict(max_win_num=max_win, max_voxel_num_per_win=wx * wy * wz, sparse_shape=list(g), win_shape=list(win), shift_list=[0, 0, 0], max_pillars_num=MP)
        wp = O.window_partition(coords.view(np.uint32) if coords.dtype != np.uint32 else coords, P, wcfg)
        gs = O.get_set(wp["gidx"], wp["cinw"], wp["vcnt"], wp["W"], dict(max_win_num=max_sets, max_voxel_num_per_win=wx * wy * wz, voxel_num_set=36, win_shape=list(win)))
        c2d = wp["c2d"][:P].astype(np.float32)                                 # (z, y, x) inside the window
        xyz = np.stack([c2d[:, 2] - wx / 2, c2d[:, 1] - wy / 2, c2d[:, 0] - wz / 2], 1).astype(np.float32)
        pe = {}
        for l in range(2):
            full = np.zeros((MP, 192), np.float32)
            full[:P] = np_(posembed(T(xyz), w, f"module.backbone_3d.input_layer.posembed_layers.{s_}.0.{l}.position_embedding_head"))
            pe[(0, l)] = full
        xin = np.zeros((MP, 192), np.float32); xin[:P] = x[:P]
        x = dsvt_blocks(dict(vfeat=xin, P=P, gss=[gs, gs], pe=pe), w, cfg, nblocks=1, stage=s_)
        if trace is not None:
            trace[("block", s_)] = (x[:P].copy(), dict(S=gs["S"], W=wp["W"], inds=gs["inds"], c2d=wp["c2d"]))
        if s_ < len(strides):
            coords2, table, parent = pool_partition(coords, P, g, strides[s_])
            x = stage_reduction_att(x[:P], table, w, f"module.backbone_3d.stage_{s_}_reduction")
            if trace is not None:
                trace[("pool", s_)] = (x.copy(), coords2.copy(), table.copy(), parent.copy())
            P = len(coords2)
            coords = np.zeros((MP, 4), np.int32); coords[:P] = coords2
            sx, sy, sz = strides[s_]
            g = (-(-g[0] // sx), -(-g[1] // sy), -(-g[2] // sz))
    return x[:P], coords[:P]
