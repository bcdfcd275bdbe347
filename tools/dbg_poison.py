"""uninitialised-read sanitiser: fill every plugin's outputs + workspace with a byte pattern before a forward; the boxes must not depend on it.
usage: dbg_poison.py [frames] [mode] [seed0]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
pkg = G.load_package(); P = pkg.plugin
FB = int(sys.argv[1]) if len(sys.argv) > 1 else 4
MODE = sys.argv[2] if len(sys.argv) > 2 else "f16"
SEED0 = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device("cuda:0")
ALL = []
_init = P.Plugin.__init__
def init(self, *a, **k):
    _init(self, *a, **k); ALL.append(self)
P.Plugin.__init__ = init
caps = pkg.pipeline.Caps() if FB == 1 else pkg.pipeline.Caps.for_frames(FB)
w = pkg.synth.make_weights()
kw = dict(linear_compute=P.COMPUTE_SPLIT) if MODE == "split" else dict(linear_compute=P.COMPUTE_F16, head_dtype=torch.float16)
pipe = pkg.pipeline.DsvtPipeline(w, caps=caps, device=dev, device_nms=True, frames=FB, **kw)
buf = np.zeros((1, FB * caps.N, 4), np.float32); ns = []
for f in range(FB):
    p = pkg.synth.lidar_like(180000, SEED0 + f); buf[0, f * caps.N:f * caps.N + len(p)] = p; ns.append(len(p))
pts, n = torch.from_numpy(buf).to(dev), torch.tensor(ns, dtype=torch.int32, device=dev)
b0, c0 = [t.clone() for t in pipe.forward(pts, n)]
torch.cuda.synchronize()
print("plugins", len(ALL), "cnt", c0.tolist())
def poison(plugs, kind, what):
    g = torch.Generator(device=dev); g.manual_seed(1234)
    for pl in plugs:
        for outs, ws in pl._cache.values():
            for t in (list(outs) if "o" in what else []) + ([ws] if "w" in what else []):
                v = t.view(-1).view(torch.uint8)
                if kind == "ff": v.fill_(255)
                elif kind == "rand": v.copy_(torch.randint(0, 256, v.shape, dtype=torch.uint8, device=dev, generator=g))
                else: v.zero_()
def same(plugs, kind, what):
    poison(plugs, kind, what)
    b, c = pipe.forward(pts, n); torch.cuda.synchronize()
    return bool(torch.equal(b, b0) and torch.equal(c, c0)), c.tolist()
for what in ("w", "o", "ow"):
    for kind in ("zero", "ff", "rand"):
        ok, c = same(ALL, kind, what)
        print(f"poison {what:2s} {kind:5s}: boxes {'same' if ok else 'DIFFER'} cnt {c}")
        if not ok:
            for i, pl in enumerate(ALL):
                poison(ALL, "zero", "ow"); pipe.forward(pts, n)
                ok1, c1 = same([pl], kind, what)
                if not ok1:
                    print("   culprit", i, pl.plugin_type, getattr(pl, "layer_name", ""), c1)
poison(ALL, "zero", "ow")
