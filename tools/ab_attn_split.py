"""Times the split-precision set attention (DsvtSetAttentionPlugin, io fp32, three fp16 products) on the bench frame's set layout, FB frames per launch:
    [DSVT_HIP_LIB=...] python tools/ab_attn_split.py [FB] [launches]      (A/B of builds: tools/build_variant.sh + DSVT_HIP_LIB)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
pkg = G.load_package(); P = pkg.plugin
dev = torch.device("cuda:0")
FB = int(sys.argv[1]) if len(sys.argv) > 1 else 4
NL = int(sys.argv[2]) if len(sys.argv) > 2 else 50
w = pkg.synth.make_weights(with_bev=False)
caps = pkg.pipeline.Caps() if FB == 1 else pkg.pipeline.Caps.for_frames(FB)
pipe = pkg.pipeline.DsvtPipeline(w, caps=caps, with_head=False, linear_compute=P.COMPUTE_SPLIT, frames=FB)
buf = np.zeros((1, FB * caps.N, 4), np.float32); ns = []
for f in range(FB):
    p = pkg.synth.lidar_like(180000, f); buf[0, f * caps.N:f * caps.N + len(p)] = p; ns.append(len(p))
st = pipe.voxel_stage(torch.from_numpy(buf).to(dev), torch.tensor(ns, dtype=torch.int32, device=dev))
for win in (0, 1):
    inds, mask, S = st["gss"][win][0], st["gss"][win][1], st["gss"][win][2]
    rows = pipe.layers[(win, 0)]["attn"]
    g = torch.Generator(device="cpu").manual_seed(win)
    qkv = torch.randn((1, FB * caps.P if FB > 1 else caps.P, 576), generator=g).to(dev)
    out = rows(qkv, inds, mask, S)[0]
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(NL):
        rows(qkv, inds, mask, S)
    e1.record(); torch.cuda.synchronize()
    print(f"split attention, window config {win}, {FB} frame(s), sets {int(S.sum()) if S.numel() > 1 else int(S)}: {e0.elapsed_time(e1) / NL * 1e3:.1f} us per launch  "
          f"checksum {float(out.double().abs().sum()):.6f}  lib={os.path.basename(os.environ.get('DSVT_HIP_LIB', 'libdsvt_hip.so'))}")
