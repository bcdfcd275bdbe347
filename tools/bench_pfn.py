"""Time DsvtPillarFeatureNetPlugin on the bench cloud (DSVT_PFN_DBG = timing ablations)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
pkg = G.load_package(); P = pkg.plugin
dev = torch.device("cuda:0")
caps = pkg.pipeline.Caps()
w = pkg.synth.make_weights(with_bev=False)
pipe = pkg.pipeline.DsvtPipeline(w, caps=caps, device=dev, linear_compute=P.COMPUTE_F16, head_dtype=torch.float16, with_head=False)
p = pkg.synth.lidar_like(180000, seed=0)
buf = np.zeros((1, caps.N, 4), np.float32); buf[0, :p.shape[0]] = p
pts = torch.from_numpy(buf).to(dev); n = torch.tensor([p.shape[0]], dtype=torch.int32, device=dev)
feat, pidx, coords, pcnt, Pn, Nk = pipe.voxelizer(pts, n)
for _ in range(3): pipe.pfn(feat, pidx, pcnt, Pn)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): pipe.pfn(feat, pidx, pcnt, Pn)
e1.record(); torch.cuda.synchronize()
print(f"pfn {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call")
