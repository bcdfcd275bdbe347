"""Graph-replay time of the four-frame fp32-grade forward with and without the shared convolution on three fp16 products (DsvtPipeline(head_mx_exclude=("shared",)):
the variant whose yaw stays below 1e-3 on all 24 clouds of tools/mx_box_sweep.py).  python tools/time_head_exclude.py"""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
pkg = G.load_package(); P = pkg.plugin
dev = torch.device("cuda:0")
FB = 4
caps = pkg.pipeline.Caps.for_frames(FB)
w = pkg.synth.make_weights()
buf = np.zeros((1, FB * caps.N, 4), np.float32); ns = []
for f in range(FB):
    p = pkg.synth.lidar_like(180000, f); buf[0, f * caps.N:f * caps.N + len(p)] = p; ns.append(len(p))
pts, n = torch.from_numpy(buf).to(dev), torch.tensor(ns, dtype=torch.int32, device=dev)
for exc in ((), ("shared",)):
    pipe = pkg.pipeline.DsvtPipeline(w, caps=caps, device=dev, device_nms=True, frames=FB, linear_compute=P.COMPUTE_SPLIT, head_mx_exclude=exc)
    pipe.capture(pts, n)
    for _ in range(3): pipe.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): pipe.replay()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 20
    print("exclude", exc, f"{t*1e3:.3f} ms per four-frame forward")
