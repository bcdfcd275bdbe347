"""Multi-stage 3-D voxel DSVT backbone on the HIP plugins (BASELINE configs[4]).

The reference has NO voxel path (its voxel z index is forced to 0: plugins/src/points2Features.cu:689-690,755) and one stage only
(src/dsvt-ai-trt.cpp:653-756), so this module has no reference counterpart: it carries the reference's own building blocks -- the z-aware voxelizer,
WindowPartition / GetSet (which carry z generically: windowPartition.cu:294-301, getSet.cu:386,461), the encoder layer -- through the multi-stage
layout of upstream DSVT's 3-D backbone: per stage its own window shape (`set_info` / `window_shape` per stage), one DSVT block (two encoder layers, sort
axes 0 / 1, position-embedding MLP over the (x, y, z) in-window coordinates evaluated once as a table over the window's cells), and between stages the
attention-style stage reduction (csrc/voxel_pool.hip).  PARITY UNPINNED beyond what the oracle can pin (indices, sets, pooling tables: bit-exact against
oracle/; features against oracle/dense_ref.py's restatement of the same published semantics).

  points [1, N, 4], n [1]
    Points2Features (grid 468 x 468 x GZ) -> fused pillar feature net (per-voxel max of the point MLP)
    stage s = 0 ..:
        WindowPartition(window_s) -> GetSet(36)
        2 x [ QKV linear with the stage's position table | set attention | out-proj + LN, FFN + LN + LN (+ block LN) ]
        s < last: DsvtVoxelPool(stride_s) -> DsvtPoolGather -> Q (pooled rows) / K / V (input rows) linears -> DsvtPoolAttentionCore -> out-proj + residual + LayerNorm
Every op is enqueued on the current stream; counts stay on the device.  Arithmetic: the fp32-grade split precision of the pillar frame ((hi, lo) fp16 operand
pairs, three MFMAs per product) in every GEMM but the stage reduction's out-projection, whose residual + LayerNorm epilogue lives on the exact-fp32 linear."""
import math
import numpy as np
import torch

from . import plugin as P
from .pipeline import fold_linear_bn

C, H, C_FFN, L_SET = 192, 8, 384, 36
X_MIN, X_MAX, Y_MIN, Y_MAX, Z_MIN, Z_MAX = -74.88, 74.88, -74.88, 74.88, -5.0, 3.0


class Dsvt3dBackbone:
    def __init__(self, weights, grid=(468, 468, 32), voxel_size=(0.32, 0.32, 0.25), windows=((12, 12, 32), (12, 12, 8)), strides=((1, 1, 4),),
                 max_points=327680, max_voxels=98304, max_win=2048, max_sets=4096, device="cuda:0", reduction_eps=1e-5):
        assert len(strides) == len(windows) - 1
        self.w, self.device = weights, torch.device(device)
        self.grid, self.windows, self.strides = tuple(grid), [tuple(w_) for w_ in windows], [tuple(s_) for s_ in strides]
        self.N, self.MP, self.W, self.S = max_points, max_voxels, max_win, max_sets
        w = weights
        ct = P.COMPUTE_SPLIT
        MP = max_voxels
        self.voxelizer = P.add_voxel_generator(max_points, max_points, MP, 4, 10, 48, X_MIN, X_MAX, Y_MIN, Y_MAX, Z_MIN, Z_MAX, *voxel_size, *grid)
        W0, b0 = fold_linear_bn(w, "module.vfe.pfn_layers.0.linear", "module.vfe.pfn_layers.0.norm", 1e-5)
        W1, b1 = fold_linear_bn(w, "module.vfe.pfn_layers.1.linear", "module.vfe.pfn_layers.1.norm", 1e-5)
        self.pfn = P.add_pillar_feature_net_op(MP, W0, b0, W1, b1, split_precision=True)
        scale = np.float32(math.sqrt(C / H))
        self.stages = []
        g = self.grid
        for s_, win in enumerate(self.windows):
            wx, wy, wz = win
            st = dict(grid=g, win=win)
            st["wp"] = P.add_window_partition(max_win, wx * wy * wz, *g, *win, 0, 0, 0)
            st["gs"] = P.add_get_set_op(max_win, wx * wy * wz, L_SET, *win, max_set_num=max_sets)
            ncell = wx * wy * wz
            cnt = torch.tensor([ncell], dtype=torch.int32, device=self.device)
            zz, yy, xx = torch.meshgrid(torch.arange(wz), torch.arange(wy), torch.arange(wx), indexing="ij")      # table row (z wy + y) wx + x
            cells = torch.stack([xx.reshape(-1).float() - wx / 2, yy.reshape(-1).float() - wy / 2, zz.reshape(-1).float() - wz / 2], 1)[None].contiguous().to(self.device)
            st["tables"], st["layers"] = [], []
            for l in range(2):
                pre = f"module.backbone_3d.input_layer.posembed_layers.{s_}.0.{l}.position_embedding_head"
                Wa, ba = fold_linear_bn(w, pre + ".0", pre + ".1", 1e-5, bias=True)
                h1 = P.add_linear_op(Wa, ba, ncell, activation=P.ACT_RELU)(cells, cnt)[0]
                st["tables"].append(P.add_linear_op(w[pre + ".3.weight"], w[pre + ".3.bias"], ncell)(h1, cnt)[0].clone())
                lp = f"module.backbone_3d.stage_{s_}.0.encoder_list.{l}"
                wi = w[lp + ".win_attn.self_attn.in_proj_weight"].copy(); bi = w[lp + ".win_attn.self_attn.in_proj_bias"].copy()
                wi[:C] /= scale; bi[:C] /= scale
                ln = lambda n: (w[lp + n + ".weight"], w[lp + n + ".bias"])
                lns = [ln(".win_attn.norm1"), ln(".win_attn.norm2"), ln(".norm")]
                if l == 1:
                    lns.append((w[f"module.backbone_3d.residual_norm_stage_{s_}.0.weight"], w[f"module.backbone_3d.residual_norm_stage_{s_}.0.bias"]))
                mk = lambda k: w[lp + k]
                mlp_w = (mk(".win_attn.self_attn.out_proj.weight"), mk(".win_attn.self_attn.out_proj.bias"), mk(".win_attn.linear1.weight"),
                         mk(".win_attn.linear1.bias"), mk(".win_attn.linear2.weight"), mk(".win_attn.linear2.bias"))
                st["layers"].append(dict(qkv=P.add_linear_op(wi, bi, MP, add_cols=2 * C, compute_type=ct, add_gather_width=wx, add_gather_height=wy),
                                         attn=P.add_set_attention_op(max_sets, L_SET, C, H, l, MP, split_precision=True),
                                         mlp=P.add_encoder_mlp_op(*mlp_w, lns, MP, split_precision=True)))
            torch.cuda.synchronize(self.device)
            if s_ < len(self.strides):
                sx, sy, sz = self.strides[s_]
                pv = sx * sy * sz
                rp = f"module.backbone_3d.stage_{s_}_reduction"
                wi = w[rp + ".self_attn.in_proj_weight"].copy(); bi = w[rp + ".self_attn.in_proj_bias"].copy()
                wi[:C] /= scale; bi[:C] /= scale                                          # nn.MultiheadAttention scales q after the bias
                st["pool"] = P.add_voxel_pool_op(MP, MP, g, (sx, sy, sz))
                st["gather"] = P.add_pool_gather_op(MP, pv, C, w[rp + ".pos_embedding"])
                st["q"] = P.add_linear_op(wi[:C], bi[:C], MP, compute_type=ct)
                st["k"] = P.add_linear_op(wi[C:2 * C], bi[C:2 * C], MP, compute_type=ct)       # (K and V per INPUT voxel: empty slots are masked out of the softmax anyway)
                st["v"] = P.add_linear_op(wi[2 * C:], bi[2 * C:], MP, compute_type=ct)
                st["core"] = P.add_pool_attention_core_op(MP, pv, C, H)
                st["o"] = P.add_linear_op(w[rp + ".self_attn.out_proj.weight"], w[rp + ".self_attn.out_proj.bias"], MP,      # (exact fp32: the LayerNorm epilogue)
                                          layer_norms=[(w[rp + ".norm.weight"], w[rp + ".norm.bias"])], ln_eps=reduction_eps)
                g = (-(-g[0] // sx), -(-g[1] // sy), -(-g[2] // sz))
            self.stages.append(st)

    def block(self, st, x, coords, Pn):
        wp = st["wp"](coords, Pn)
        gs = st["gs"](wp[0], wp[1], wp[2], wp[3])
        c2d, inds, mask, S = wp[4], gs[0], gs[1], gs[2]
        xb = x
        for l, L in enumerate(st["layers"]):
            qkv = L["qkv"](x, Pn, st["tables"][l], c2d)[0]
            att = L["attn"](qkv, inds, mask, S)[0]
            x = (L["mlp"](att, Pn, x, xb) if l == 1 else L["mlp"](att, Pn, x))[0]
        return x, dict(c2d=c2d, inds=inds, mask=mask, S=S, W=wp[3])

    def reduce(self, st, x, coords, Pn):
        coords2, table, parent, P2, rows = st["pool"](coords, Pn)
        src, kin = st["gather"](x, table, P2)
        q = st["q"](src, P2)[0]
        k = st["k"](kin, Pn)[0]
        v = st["v"](x, Pn)[0]
        ctx = st["core"](q, k, v, table, P2)[0]
        y = st["o"](ctx, P2, src)[0]
        return y, coords2, P2, dict(table=table, parent=parent)

    def forward(self, points, n, trace=None):
        """points [1, max_points, 4], n [1] -> (features [1, max_voxels, 192] of the LAST stage's voxels, their coords [1, max_voxels, 4] (b, z, y, x), count [1])"""
        feat, pidx, coords, pcnt, Pn, _Nk = self.voxelizer(points, n)
        x = self.pfn(feat, pidx, pcnt, Pn)[0]
        for s_, st in enumerate(self.stages):
            if trace is not None:
                trace[("in", s_)] = (x.clone(), coords.clone(), Pn.clone())
            x, info = self.block(st, x, coords, Pn)
            if trace is not None:
                trace[("block", s_)] = (x.clone(), {k_: v_.clone() for k_, v_ in info.items()})
            if "pool" in st:
                x, coords, Pn, pinfo = self.reduce(st, x, coords, Pn)
                if trace is not None:
                    trace[("pool", s_)] = (x.clone(), coords.clone(), Pn.clone(), {k_: v_.clone() for k_, v_ in pinfo.items()})
        return x, coords, Pn
