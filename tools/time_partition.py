"""the fused set partition alone (sp_count / sp_scan / sp_scatter / sp_window): GPU time per launch (events) at FB frames per launch; under
`rocprofv3 --kernel-trace --stats` the per-kernel split.  usage: python tools/time_partition.py [frames]"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
pkg = G.load_package(); P = pkg.plugin
FB = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
caps = pkg.pipeline.Caps() if FB == 1 else pkg.pipeline.Caps.for_frames(FB)
pipe = pkg.pipeline.DsvtPipeline(pkg.synth.make_weights(), caps=caps, device=dev, device_nms=True, frames=FB, linear_compute=P.COMPUTE_F16, head_dtype=torch.float16)
buf = np.zeros((1, FB * caps.N, 4), np.float32); ns = []
for f in range(FB):
    p = pkg.synth.lidar_like(180000, f); buf[0, f * caps.N:f * caps.N + len(p)] = p; ns.append(len(p))
pts, n = torch.from_numpy(buf).to(dev), torch.tensor(ns, dtype=torch.int32, device=dev)
feat, pidx, coords, pcnt, Pn, Nk = pipe.voxelizer(pts, n)
iters = 30
for _ in range(3):
    pipe.part(coords, Pn)
torch.cuda.synchronize()
e = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
e[0].record()
for i in range(iters):
    pipe.part(coords, Pn); e[i + 1].record()
torch.cuda.synchronize()
ts = sorted(e[i].elapsed_time(e[i + 1]) * 1e3 for i in range(iters))
print(f"set partition, {FB} frames per launch, {int(Pn[0])} pillars: median {ts[len(ts) // 2]:.1f} us, min {ts[0]:.1f} us per launch (host-launched, back to back)")
