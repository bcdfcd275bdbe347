"""DSVT_PFN_TRACE=1 with the ablate build: clock64 stamps of workgroup 0's first groups (mark order per group: group start, tables +
starts ready, units ready, point pass done, barrier passed).  usage: DSVT_HIP_LIB=.../libdsvt_hip_ablate.so DSVT_PFN_TRACE=1 python tools/trace_pfn.py [frames]"""
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
st = pipe.voxel_stage(torch.from_numpy(buf).to(dev), torch.tensor(ns, dtype=torch.int32, device=dev))
torch.cuda.synchronize()
