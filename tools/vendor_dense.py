"""The dense BEV stage on PyTorch-ROCm / MIOpen (round 1's glue), kept OUTSIDE the product package for error attribution
(tools/err_attrib.py) and stage timing (tools/stage_times.py): `VendorDensePipeline` is a DsvtPipeline whose head() runs
F.conv2d / F.conv_transpose2d / torch.topk instead of DsvtConv2dPlugin / CenterHeadTopKPlugin.  The product package
(dsvt-ai-trt_amd/) holds no vendor-library stage."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402

pkg = G.load_package()
from dsvt_ai_trt_amd.pipeline import DsvtPipeline, bn_fold, GX, TOP_K  # noqa: E402


class VendorDensePipeline(DsvtPipeline):
    def __init__(self, weights, **kw):
        super().__init__(weights, **kw)
        self._build_dense(weights)

    def head(self, x, st):
        src = x if self.head_dtype == torch.float32 else (self._xh if self.f16 else x.to(torch.float16))
        if not hasattr(self, "_m2b"):
            self._m2b = pkg.plugin.add_map_2_bev_op(self.caps.P, 192, 468, 468)
        bev = self._m2b(src, st["coords"], st["P"])[0].permute(0, 3, 1, 2)     # NCHW view of channels-last memory (:1131-1133)
        if bev.dtype != self.head_dtype:
            bev = bev.to(self.head_dtype)
        return self._post(self.filter(*self._decode(self._bev(bev))))

    # ---- dense glue: BN folded into the convolutions, channels-last ----------
    def _conv_params(self, w, conv, bn):
        s, sh = bn_fold(w, bn, 1e-3)                                                               # :191,208,239
        W = torch.from_numpy(w[conv + ".weight"] * s[:, None, None, None])
        return (W.to(self.device, self.head_dtype).contiguous(memory_format=torch.channels_last),
                torch.from_numpy(sh).to(self.device, self.head_dtype))

    def _build_dense(self, w):
        d = self.dense = {}
        for (i, nb) in ((0, 2), (1, 3), (2, 3)):
            for j in range(nb):
                p = f"module.backbone_2d.blocks.{i}.{j}"
                d[p + ".1"] = self._conv_params(w, p + ".conv1", p + ".bn1")
                d[p + ".2"] = self._conv_params(w, p + ".conv2", p + ".bn2")
                if j == 0:
                    d[p + ".d"] = self._conv_params(w, p + ".downsample_layer.0", p + ".downsample_layer.1")
        for i in range(3):
            p = f"module.backbone_2d.deblocks.{i}"
            s, sh = bn_fold(w, p + ".1", 1e-3)
            W = torch.from_numpy(w[p + ".0.weight"] * s[None, :, None, None])                       # ConvTranspose [in,out,k,k]
            d[p] = (W.to(self.device, self.head_dtype), torch.from_numpy(sh).to(self.device, self.head_dtype))
        d["shared"] = self._conv_params(w, "module.dense_head.shared_conv.0", "module.dense_head.shared_conv.1")
        # the five live heads' first convs share their input: one 64 -> 320 convolution (iou head is dead, :1440-1452)
        names = ["center", "center_z", "dim", "rot", "hm"]
        Ws, bs = zip(*[self._conv_params(w, f"module.dense_head.heads_list.0.{n}.0.0", f"module.dense_head.heads_list.0.{n}.0.1")
                       for n in names])
        d["heads0"] = (torch.cat(Ws, 0).contiguous(memory_format=torch.channels_last), torch.cat(bs, 0))
        outs = [2, 1, 3, 2, 10]
        W2 = torch.zeros((sum(outs), 64 * 5, 3, 3), dtype=torch.float32)
        b2 = torch.zeros((sum(outs),), dtype=torch.float32)
        o = 0
        for k, (n, no) in enumerate(zip(names, outs)):                                              # block-diagonal second convs
            W2[o:o + no, 64 * k:64 * (k + 1)] = torch.from_numpy(w[f"module.dense_head.heads_list.0.{n}.1.weight"])
            b2[o:o + no] = torch.from_numpy(w[f"module.dense_head.heads_list.0.{n}.1.bias"])
            o += no
        d["heads1"] = (W2.to(self.device, self.head_dtype).contiguous(memory_format=torch.channels_last),
                       b2.to(self.device, self.head_dtype))


    def _decode_nhwc(self, o):
        """same as _decode for an NHWC [1,H,W,18] head output"""
        of = o.reshape(-1, 18)
        hm = torch.sigmoid(of[:, 8:18].t().contiguous())                # [10, H*W]
        sc1, idx1 = torch.topk(hm, TOP_K, dim=1)
        sc2, idx2 = torch.topk(sc1.reshape(-1), TOP_K)
        cls = (idx2 // TOP_K).to(torch.int32)
        ind = idx1.reshape(-1)[idx2]
        ys, xs = (ind // GX).to(torch.int32), (ind % GX).to(torch.int32)
        g = of[ind]                                                     # [K, 18]
        center = g[:, 0:2].contiguous(); center_z = g[:, 2:3].contiguous()
        dim = torch.exp(g[:, 3:6]).contiguous()
        angle = torch.atan(g[:, 7:8] / g[:, 6:7]).contiguous()
        return (sc2.reshape(1, -1), cls.reshape(1, -1), xs.reshape(1, -1), ys.reshape(1, -1), center.reshape(1, 1, -1, 2),
                center_z.reshape(1, 1, -1, 1), angle.reshape(1, 1, -1, 1), dim.reshape(1, 1, -1, 3))

    def _bev(self, x):
        d = self.dense
        ups = []
        for (i, stride, nb, k) in ((0, 1, 2, 1), (1, 2, 3, 2), (2, 2, 3, 4)):
            for j in range(nb):
                p = f"module.backbone_2d.blocks.{i}.{j}"
                s = stride if j == 0 else 1
                y = F.relu(F.conv2d(x, *d[p + ".1"], stride=s, padding=1))
                y = F.conv2d(y, *d[p + ".2"], stride=1, padding=1)
                idn = F.conv2d(x, *d[p + ".d"], stride=s) if j == 0 else x
                x = F.relu(y + idn)
            Wd, bd = d[f"module.backbone_2d.deblocks.{i}"]
            ups.append(F.relu(F.conv_transpose2d(x, Wd, bd, stride=k)))
        f = torch.cat(ups, 1)
        sh = F.relu(F.conv2d(f, *d["shared"], padding=1))
        h0 = F.relu(F.conv2d(sh, *d["heads0"], padding=1))
        return F.conv2d(h0, *d["heads1"], padding=1).float()       # [1, 18, 468, 468]: center2 cz1 dim3 rot2 hm10

    def _decode(self, o):
        """sigmoid / exp / two-stage top-K / gathers / atan(sin/cos)  (src/dsvt-ai-trt.cpp:1479-1669)"""
        o = o[0]
        hm = torch.sigmoid(o[8:18]).reshape(10, -1)
        sc1, idx1 = torch.topk(hm, TOP_K, dim=1)
        sc2, idx2 = torch.topk(sc1.reshape(-1), TOP_K)
        cls = (idx2 // TOP_K).to(torch.int32)
        ind = idx1.reshape(-1)[idx2]
        ys, xs = (ind // GX).to(torch.int32), (ind % GX).to(torch.int32)
        g = o.reshape(18, -1)[:, ind]                               # [18, K]
        center = g[0:2].T.contiguous(); center_z = g[2:3].T.contiguous()
        dim = torch.exp(g[3:6]).T.contiguous()
        angle = torch.atan(g[7:8] / g[6:7]).T.contiguous()          # rot[1]/rot[0]: sin/cos slices :1494-1501
        return (sc2.reshape(1, -1), cls.reshape(1, -1), xs.reshape(1, -1), ys.reshape(1, -1), center.reshape(1, 1, -1, 2),
                center_z.reshape(1, 1, -1, 1), angle.reshape(1, 1, -1, 1), dim.reshape(1, 1, -1, 3))

