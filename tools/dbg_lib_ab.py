"""Dump the split-precision frame's intermediate tensors (PFN output, x after every encoder layer, the first layer's QKV / attention) to an .npz -- run once per library
(DSVT_HIP_LIB=...) and compare the files: which kernel's bits changed between two builds.  python tools/dbg_lib_ab.py out.npz"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
pkg = G.load_package(); P = pkg.plugin
dev = torch.device("cuda:0")
caps = pkg.pipeline.Caps()
pipe = pkg.pipeline.DsvtPipeline(pkg.synth.make_weights(), caps=caps, device=dev, linear_compute=P.COMPUTE_SPLIT, with_head=False)
p = pkg.synth.lidar_like(180000, 0); buf = np.zeros((1, caps.N, 4), np.float32); buf[0, :len(p)] = p
pts, n = torch.from_numpy(buf).to(dev), torch.tensor([len(p)], dtype=torch.int32, device=dev)
st = pipe.voxel_stage(pts, n)
out = {"vfeat": st["vfeat"].cpu().numpy()}
L = pipe.layers[(0, 0)]
qkv = L["qkv"](st["vfeat"], st["P"], pipe.pos_tables[(0, 0)], st["wps"][0][4])[0]
att = L["attn"](qkv, st["gss"][0][0], st["gss"][0][1], st["gss"][0][2])[0]
out["qkv00"] = qkv.cpu().numpy(); out["att00"] = att.cpu().numpy()
tr = {}
x = pipe.backbone(st, trace=tr)
for k, v in tr.items(): out[f"x{k[0]}{k[1]}"] = v.cpu().numpy()
torch.cuda.synchronize()
np.savez(sys.argv[1], **out)
print("saved", sorted(out))
