"""Time DsvtPillarFeatureNetPlugin on the bench cloud: python tools/bench_pfn.py [frames] [f16|split]
With DSVT_HIP_LIB=dsvt-ai-trt_amd/libdsvt_hip_ablate.so:  DSVT_PFN_DBG=<bits> timing ablations (1 no point gather, 2 no layer-0 MFMA, 4 no layer-1 MFMA,
8 no stores, 16 no per-pillar GEMM, 32 no m maxima);  DSVT_PFN_TRACE=1 prints s_memtime stamps of one wave, a row of seven per tile (tile start, next rows
requested, layer 0 done, operand built, m stored, layer 1 done, maxima kept) and per group (epilogue start, GEMM done, stores issued, 4 x the same)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
pkg = G.load_package(); P = pkg.plugin
dev = torch.device("cuda:0")
FB = int(sys.argv[1]) if len(sys.argv) > 1 else 1
MODE = sys.argv[2] if len(sys.argv) > 2 else "f16"
caps = pkg.pipeline.Caps() if FB == 1 else pkg.pipeline.Caps.for_frames(FB)
w = pkg.synth.make_weights(with_bev=False)
kw = dict(linear_compute=P.COMPUTE_SPLIT) if MODE == "split" else dict(linear_compute=P.COMPUTE_F16, head_dtype=torch.float16)
pipe = pkg.pipeline.DsvtPipeline(w, caps=caps, device=dev, with_head=False, frames=FB, **kw)
buf = np.zeros((1, FB * caps.N, 4), np.float32); ns = []
for f in range(FB):
    p = pkg.synth.lidar_like(180000, f); buf[0, f * caps.N:f * caps.N + len(p)] = p; ns.append(len(p))
pts, n = torch.from_numpy(buf).to(dev), torch.tensor(ns, dtype=torch.int32, device=dev)
feat, pidx, coords, pcnt, Pn, Nk = pipe.voxelizer(pts, n)[:6]
for _ in range(3): pipe.pfn(feat, pidx, pcnt, Pn)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): out = pipe.pfn(feat, pidx, pcnt, Pn)
e1.record(); torch.cuda.synchronize()
o = out[0].float()
print(f"pfn {MODE} {FB} frames: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call   (pillars {int(Pn.sum())}, checksum {float(o.double().sum()):.6f} max {float(o.max()):.6f})")
