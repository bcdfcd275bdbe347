"""Do two kernels from two streams really share the GPU?  Times kernel A alone, kernel B alone and both at once (each looped on its own
stream): wall(A || B) ~ max(A, B) means they overlap, ~ A + B means the hardware runs them one after the other.

    python tools/corun.py
"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
pkg = G.load_package(); P = pkg.plugin
dev = torch.device("cuda:0")
caps = pkg.pipeline.Caps()
w = pkg.synth.make_weights()
pipe = pkg.pipeline.DsvtPipeline(w, caps=caps, device=dev, linear_compute=P.COMPUTE_F16, head_dtype=torch.float16, device_nms=True)
p = pkg.synth.lidar_like(180000, 0)
buf = np.zeros((1, caps.N, 4), np.float32); buf[0, :p.shape[0]] = p
pts = torch.from_numpy(buf).to(dev); n = torch.tensor([p.shape[0]], dtype=torch.int32, device=dev)
pipe.forward(pts, n); torch.cuda.synchronize()
st = pipe.voxel_stage(pts, n)
Pn = st["P"]
L = pipe.layers[(0, 0)]
xh = st["vfeat16"]
qkv = L["qkv"](xh, Pn, pipe.pos_tables[(0, 0)], st["wps"][0][4])[0]
inds, mask, S = st["gss"][0]
att = L["attn"](qkv, inds, mask, S)[0]
x = st["vfeat"]
bev = torch.randn((1, 468, 468, 128), device=dev).half()
conv = pipe.hops["module.backbone_2d.blocks.0.1.1"]          # 468 x 468, 128 -> 128, 3 x 3
kern = {
    "conv128": lambda: conv(bev),
    "attention": lambda: L["attn"](qkv, inds, mask, S),
    "qkv": lambda: L["qkv"](xh, Pn, pipe.pos_tables[(0, 0)], st["wps"][0][4]),
    "mlp": lambda: L["mlp"](att, Pn, x),
    "pfn": lambda: pipe.pfn(st["feat"], st["pidx"], st["pcnt"], Pn),
}
for f in kern.values():
    f()
torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
REP = 200


def run(fa, fb):
    for _ in range(100):                 # clocks up, caches warm
        (fa or fb)()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(REP):
        if fa is not None:
            with torch.cuda.stream(s1): fa()
        if fb is not None:
            with torch.cuda.stream(s2): fb()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / REP * 1e6


alone = {k: run(f, None) for k, f in kern.items()}
print("alone (us per launch, host-launched back to back):", {k: round(v, 1) for k, v in alone.items()})
for a, b in (("conv128", "attention"), ("conv128", "mlp"), ("conv128", "qkv"), ("conv128", "pfn"), ("mlp", "mlp"), ("mlp", "attention"), ("qkv", "attention"),
             ("attention", "attention"), ("conv128", "conv128")):
    t = run(kern[a], kern[b])
    print(f"{a:10s} || {b:10s}: {t:7.1f} us per pair   sum {alone[a] + alone[b]:7.1f}   max {max(alone[a], alone[b]):7.1f}   "
          f"overlap {100 * (alone[a] + alone[b] - t) / min(alone[a], alone[b]):5.1f} % of the shorter one")
