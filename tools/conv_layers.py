"""Per-layer durations of the dense stage inside the frame (HIP events around every DsvtConv2dPlugin launch of an eager forward, real BEV maps):
    python tools/conv_layers.py [frames per forward] [mode: split | splitmx | f16]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
pkg = G.load_package(); P = pkg.plugin
dev = torch.device("cuda:0")
FB = int(sys.argv[1]) if len(sys.argv) > 1 else 4
mode = sys.argv[2] if len(sys.argv) > 2 else "split"
kw = {"split": dict(linear_compute=P.COMPUTE_SPLIT), "splitmx": dict(linear_compute=P.COMPUTE_SPLIT, head_mx=True), "f16": dict(linear_compute=P.COMPUTE_F16, head_dtype=torch.float16)}[mode]
caps = pkg.pipeline.Caps() if FB == 1 else pkg.pipeline.Caps.for_frames(FB)
w = pkg.synth.make_weights()
pipe = pkg.pipeline.DsvtPipeline(w, caps=caps, device=dev, device_nms=True, frames=FB, **kw)
buf = np.zeros((1, FB * caps.N, 4), np.float32); ns = []
for f in range(FB):
    p = pkg.synth.lidar_like(180000, f); buf[0, f * caps.N:f * caps.N + len(p)] = p; ns.append(len(p))
pts, n = torch.from_numpy(buf).to(dev), torch.tensor(ns, dtype=torch.int32, device=dev)
names = {id(op): k for k, op in (pipe.sops if getattr(pipe, "split_head", False) else pipe.hops).items()}
for _ in range(3): pipe.forward(pts, n)
torch.cuda.synchronize()
REPS = 5
acc = {}
for _ in range(REPS):
    P.PROFILE = {"DsvtConv2dPlugin": []}
    pipe.forward(pts, n); torch.cuda.synchronize()
    for j, (e0, e1, pl) in enumerate(P.PROFILE["DsvtConv2dPlugin"]):
        acc.setdefault(j, [pl, 0.0])[1] += e0.elapsed_time(e1) * 1e3 / REPS
    P.PROFILE = None
tot = 0.0
for j, (pl, us) in acc.items():
    f = pl.fields
    Ho = (f["in_height"] + 2 * f["padding"] - f["kernel_size"]) // f["stride"] + 1
    cin = f["in_channels"] // 3 if getattr(pl, "split_in", False) else f["in_channels"]
    fl = FB * 2.0 * Ho * Ho * f.get("pixel_shuffle", 1) ** 2 * f["out_channels"] * f["kernel_size"] ** 2 * cin
    tot += us
    print(f"{names.get(id(pl), '?'):42s} {f['in_height']:4d}^2 {cin:4d}->{f['out_channels']:4d} k{f['kernel_size']} s{f['stride']} up{f.get('pixel_shuffle', 1)}  {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s of products")
print(f"dense stage, {FB} frame(s) per forward, mode {mode}: {tot / 1e3:.3f} ms")
