"""BASELINE configs[4]: lidar_like(300000, seed) through the two-stage 3-D voxel backbone (pipeline3d.Dsvt3dBackbone): eager and HIP-graph-replay time per cloud,
and HIP events around every plugin launch of one forward (where the time goes).   python tools/time_backbone3d.py [clouds]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
pkg = G.load_package(); P = pkg.plugin
dev = torch.device("cuda:0")
NC = int(sys.argv[1]) if len(sys.argv) > 1 else 4
net = pkg.pipeline3d.Dsvt3dBackbone(pkg.synth.make_weights_3d(), device=dev)
clouds = []
for s in range(NC):
    p = pkg.synth.lidar_like(300000, s)
    buf = np.zeros((1, net.N, 4), np.float32); buf[0, :len(p)] = p
    clouds.append((torch.from_numpy(buf).to(dev), torch.tensor([len(p)], dtype=torch.int32, device=dev)))
for pts, n in clouds[:2]:
    x, coords, Pn = net.forward(pts, n)
torch.cuda.synchronize()
print("voxels of the last stage, cloud 1:", int(Pn[0]), flush=True)
t0 = time.perf_counter()
for i in range(12):
    net.forward(*clouds[i % NC])
torch.cuda.synchronize()
print(f"eager: {(time.perf_counter() - t0) / 12 * 1e3:.3f} ms per cloud")
sin = (torch.zeros_like(clouds[0][0]), torch.zeros_like(clouds[0][1]))
sin[0].copy_(clouds[0][0]); sin[1].copy_(clouds[0][1])
for _ in range(3):                     # (warm-up with the STATIC inputs on the current stream, as DsvtPipeline.capture does: every buffer exists before the capture)
    net.forward(*sin)
torch.cuda.synchronize()
print("capturing", flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = net.forward(*sin)
for _ in range(3): g.replay()
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(24):
    sin[0].copy_(clouds[i % NC][0]); sin[1].copy_(clouds[i % NC][1]); g.replay()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 24 * 1e3
print(f"graph replay: {ms:.3f} ms per cloud = {1e3 / ms:.1f} clouds/s (300k points, 468 x 468 x 32 grid, two stages)")
P.PROFILE = {k: [] for k in P.plugin_types()}
net.forward(*clouds[0]); torch.cuda.synchronize()
rows = sorted(((sum(e0.elapsed_time(e1) for e0, e1, _ in v) * 1e3, len(v), k) for k, v in P.PROFILE.items() if v), reverse=True)
P.PROFILE = None
for us, cnt, k in rows:
    print(f"  {k:32s} {cnt:3d} launches {us:9.1f} us")
