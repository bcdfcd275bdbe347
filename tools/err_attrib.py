"""Where does the fp16 frame's box error come from?  Runs the fp32 HIP pipeline (which matches the oracle to 1e-5) and the
fp16 pipeline on the same frame and swaps stages between them: the head tensor of each hybrid is compared with the all-fp32
head tensor at the 500 candidate cells.

    python tools/err_attrib.py [--frame 000000 | --points 180000]
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frame", default="000000")
    ap.add_argument("--points", type=int, default=0)
    args = ap.parse_args()
    pkg = G.load_package()
    P = pkg.plugin
    from tests import cases
    dev = "cuda:0"
    w = pkg.synth.make_weights()
    if args.points:
        caps = pkg.pipeline.Caps()
        p = pkg.synth.lidar_like(args.points, 0)
        pts = np.zeros((caps.N, 4), np.float32); pts[:p.shape[0]] = p; n = p.shape[0]
    else:
        caps = pkg.pipeline.Caps.reference()
        pts, n = cases.load_frame(args.frame, caps.N)
    d_pts = torch.from_numpy(pts[None]).to(dev); d_n = torch.tensor([n], dtype=torch.int32, device=dev)
    from tools.vendor_dense import VendorDensePipeline
    p32 = VendorDensePipeline(w, caps=caps, device=dev)      # PyTorch fp32 dense stage (tools/vendor_dense.py): its intermediates are needed
    p16 = pkg.pipeline.DsvtPipeline(w, caps=caps, device=dev, linear_compute=P.COMPUTE_F16, head_dtype=torch.float16)

    # ---- all-fp32 reference with its intermediate tensors -----------------------------------------
    p32.forward(d_pts, d_n)                                   # (creates the fp32 Map2Bev op of the vendor head)
    st32 = p32.voxel_stage(d_pts, d_n)
    x32 = p32.backbone(st32).clone()
    Pn = int(st32["P"][0])

    def dense32(x, upto=None):
        """pipe32's dense stage on a [1,maxP,192] fp32 voxel tensor; returns dict of NCHW intermediates"""
        bev = p32._m2b(x, st32["coords"], st32["P"])[0].permute(0, 3, 1, 2)
        d = p32.dense
        out = {}
        xx = bev
        ups = []
        for (i, stride, nb, k) in ((0, 1, 2, 1), (1, 2, 3, 2), (2, 2, 3, 4)):
            for j in range(nb):
                p = f"module.backbone_2d.blocks.{i}.{j}"
                s = stride if j == 0 else 1
                y = F.relu(F.conv2d(xx, *d[p + ".1"], stride=s, padding=1))
                y = F.conv2d(y, *d[p + ".2"], stride=1, padding=1)
                idn = F.conv2d(xx, *d[p + ".d"], stride=s) if j == 0 else xx
                xx = F.relu(y + idn)
            Wd, bd = d[f"module.backbone_2d.deblocks.{i}"]
            ups.append(F.relu(F.conv_transpose2d(xx, Wd, bd, stride=k)))
        out["cat"] = torch.cat(ups, 1)
        out["shared"] = F.relu(F.conv2d(out["cat"], *d["shared"], padding=1))
        out["h0"] = F.relu(F.conv2d(out["shared"], *d["heads0"], padding=1))
        out["head"] = F.conv2d(out["h0"], *d["heads1"], padding=1).float()
        return out

    r32 = dense32(x32)
    head32 = r32["head"][0].permute(1, 2, 0).reshape(-1, 18)          # [H*W, 18]
    hm = torch.sigmoid(head32[:, 8:18].t().contiguous())
    sc1, idx1 = torch.topk(hm, 500, dim=1)
    sc2, idx2 = torch.topk(sc1.reshape(-1), 500)
    ind = idx1.reshape(-1)[idx2]
    cls = idx2 // 500
    keep = sc2 >= 0.3
    print(f"frame: n={n} P={Pn} candidates over threshold: {int(keep.sum())}")
    ref = head32[ind]

    def report(name, head_nhwc):
        got = head_nhwc.reshape(-1, 18).float()[ind]
        dlt = (got - ref).abs()
        k = keep
        sc = (torch.sigmoid(got[torch.arange(500), 8 + cls]) - torch.sigmoid(ref[torch.arange(500), 8 + cls])).abs()
        size = (torch.exp(got[:, 3:6]) - torch.exp(ref[:, 3:6])).abs()
        full = (head_nhwc.reshape(-1, 18).float() - head32).abs()
        print(f"{name:58s} kept boxes: centre*0.32 {0.32 * dlt[k][:, 0:2].max().item():.2e}  z {dlt[k][:, 2].max().item():.2e}  "
              f"size {size[k].max().item():.2e}  rot {dlt[k][:, 6:8].max().item():.2e}  score {sc[k].max().item():.2e}   "
              f"| all cells: max {full.max().item():.2e} rms {full.pow(2).mean().sqrt().item():.2e}")

    nhwc16 = lambda t: t.permute(0, 2, 3, 1).contiguous().half()

    # ---- everything fp16 (the benched path) ------------------------------------------------------------
    st16 = p16.voxel_stage(d_pts, d_n)
    x16 = p16.backbone(st16).clone()
    xh16 = p16._xh.clone()
    bev16 = p16.map2bev(xh16, st16["coords"], st16["P"])[0]
    report("A. all fp16 (bench path)", p16._bev_hip(bev16))
    # ---- fp16 backbone, fp32 dense stage -------------------------------------------------------------
    report("B. fp16 voxel+DSVT blocks, fp32 dense stage", dense32(x16)["head"][0].permute(1, 2, 0))
    vf = (st16["vfeat"][0, :Pn] - st32["vfeat"][0, :Pn]).abs()
    print(f"     PFN features fp16-path vs fp32: max {vf.max().item():.2e} rms {vf.pow(2).mean().sqrt().item():.2e} (scale {st32['vfeat'][0, :Pn].abs().max().item():.2f})")
    dx = (x16[0, :Pn] - x32[0, :Pn]).abs()
    print(f"     backbone output fp16-path vs fp32: max {dx.max().item():.2e} rms {dx.pow(2).mean().sqrt().item():.2e}")
    # ---- fp32 backbone, fp16 dense stage -----------------------------------------------------------
    bev = p16.map2bev(x32.half(), st32["coords"], st32["P"])[0]
    report("C. fp32 voxel+DSVT blocks, fp16 dense stage (25 convs)", p16._bev_hip(bev))
    # ---- fp32 up to the concatenated BEV features, fp16 CenterHead (3 convs) ---------------------
    ops = p16.hops
    p16.cat_bev.copy_(nhwc16(r32["cat"]))
    sh = ops["shared"](p16.cat_bev)[0]
    report("D. fp32 up to cat_bev, fp16 shared+heads0+heads1", ops["heads1"](ops["heads0"](sh)[0])[0])
    report("E. fp32 up to shared, fp16 heads0+heads1", ops["heads1"](ops["heads0"](nhwc16(r32["shared"]))[0])[0])
    report("F. fp32 up to h0, fp16 heads1 only", ops["heads1"](nhwc16(r32["h0"]))[0])
    # weights-only rounding of the last layer: fp32 activations (h0) x fp16-rounded W
    W1, b1 = p32.dense["heads1"]
    report("G. fp32 everything, heads1 weights rounded to fp16", F.conv2d(r32["h0"], W1.half().float(), b1, padding=1)[0].permute(1, 2, 0))
    report("H. fp32 everything, h0 rounded to fp16 (fp32 W)", F.conv2d(r32["h0"].half().float(), W1, b1, padding=1)[0].permute(1, 2, 0))
    # fp16 PFN only
    st_mix = dict(st32); st_mix["vfeat"] = st16["vfeat"]
    xm = p32.backbone(st_mix).clone()
    report("I. fp16-path PFN only (fp32 rest)", dense32(xm)["head"][0].permute(1, 2, 0))
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
