"""the voxelizer alone: GPU time per launch (events) at FB frames per launch; DSVT_P2F_TRACE=1 + the ablate build prints p2f_bins' phase stamps.
usage: [DSVT_HIP_LIB=dsvt-ai-trt_amd/libdsvt_hip_ablate.so DSVT_P2F_TRACE=1] python tools/time_p2f.py [frames]"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
pkg = G.load_package(); P = pkg.plugin
FB = int(sys.argv[1]) if len(sys.argv) > 1 else 4
SEED0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda:0")
caps = pkg.pipeline.Caps() if FB == 1 else pkg.pipeline.Caps.for_frames(FB)
pipe = pkg.pipeline.DsvtPipeline(pkg.synth.make_weights(), caps=caps, device=dev, device_nms=True, frames=FB, linear_compute=P.COMPUTE_F16, head_dtype=torch.float16)
buf = np.zeros((1, FB * caps.N, 4), np.float32); ns = []
for f in range(FB):
    p = pkg.synth.lidar_like(180000, SEED0 + f); buf[0, f * caps.N:f * caps.N + len(p)] = p; ns.append(len(p))
pts, n = torch.from_numpy(buf).to(dev), torch.tensor(ns, dtype=torch.int32, device=dev)
vox = pipe.voxelizer
trace = os.environ.get("DSVT_P2F_TRACE")
iters = 1 if trace else 30
for _ in range(0 if trace else 3):
    vox(pts, n)
torch.cuda.synchronize()
e = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
e[0].record()
for i in range(iters):
    vox(pts, n); e[i + 1].record()
torch.cuda.synchronize()
ts = sorted(e[i].elapsed_time(e[i + 1]) * 1e3 for i in range(iters))
print(f"voxelizer, {FB} frames per launch: median {ts[len(ts) // 2]:.1f} us, min {ts[0]:.1f} us per launch (host-launched, back to back)")
