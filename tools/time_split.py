"""per-stage GPU time of the split-precision frame (host-launched, events): python tools/time_split.py [frames]"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
pkg = G.load_package(); P = pkg.plugin
FB = int(sys.argv[1]) if len(sys.argv) > 1 else 1
MODE = sys.argv[2] if len(sys.argv) > 2 else "split"
dev = torch.device("cuda:0")
caps = pkg.pipeline.Caps() if FB == 1 else pkg.pipeline.Caps.for_frames(FB)
w = pkg.synth.make_weights()
kw = dict(linear_compute=P.COMPUTE_SPLIT) if MODE == "split" else dict(linear_compute=P.COMPUTE_F16, head_dtype=torch.float16)
pipe = pkg.pipeline.DsvtPipeline(w, caps=caps, device=dev, device_nms=True, frames=FB, **kw)
buf = np.zeros((1, FB * caps.N, 4), np.float32); ns = []
for f in range(FB):
    p = pkg.synth.lidar_like(180000, f); buf[0, f * caps.N:f * caps.N + len(p)] = p; ns.append(len(p))
pts, n = torch.from_numpy(buf).to(dev), torch.tensor(ns, dtype=torch.int32, device=dev)
for _ in range(3):
    pipe.forward(pts, n)
torch.cuda.synchronize()
prof = {}
P.PROFILE = prof
class Any(dict):
    def get(self, k, d=None):
        return self.setdefault(k, [])
P.PROFILE = Any()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); pipe.forward(pts, n); e1.record(); torch.cuda.synchronize()
tot = 0
for k, lst in P.PROFILE.items():
    ms = sum(a.elapsed_time(b) for a, b, _ in lst); tot += ms
    print(f"{k:32s} {len(lst):3d} launches {ms:8.3f} ms  ({ms / FB:.3f} per frame)")
for a, b, pl in P.PROFILE.get("DsvtConv2dPlugin", []):
    f = pl.fields
    print(f"   conv {f['in_height']:4d}^2 {f['in_channels']:4d} -> {f['out_channels']:4d} k{f['kernel_size']} s{f['stride']} up{f.get('pixel_shuffle', 1)} res{f.get('has_residual', 0)}: {a.elapsed_time(b) * 1e3:8.1f} us")
print(f"sum {tot:.3f} ms, wall {e0.elapsed_time(e1):.3f} ms, per frame {e0.elapsed_time(e1) / FB:.3f}")
P.PROFILE = None
g_out = pipe.capture(pts, n)
torch.cuda.synchronize()
e0.record()
for _ in range(10):
    pipe.replay()
e1.record(); torch.cuda.synchronize()
print(f"graph replay: {e0.elapsed_time(e1) / 10:.3f} ms per forward, {e0.elapsed_time(e1) / 10 / FB:.3f} per frame")
