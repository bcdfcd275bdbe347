"""run as two concurrent processes: NS pipelines on NS streams, different clouds per stream, eager forwards launched back to back;
every plugin's output buffers are compared with those of iteration 0.  usage: dbg_two_proc.py rank [frames] [mode] [streams]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
pkg = G.load_package(); P = pkg.plugin
RANK = int(sys.argv[1]); FB = int(sys.argv[2]) if len(sys.argv) > 2 else 4
MODE = sys.argv[3] if len(sys.argv) > 3 else "f16"; NS = int(sys.argv[4]) if len(sys.argv) > 4 else 2
dev = torch.device("cuda:0")
ALL = []
_init = P.Plugin.__init__
def init(self, *a, **k):
    _init(self, *a, **k); ALL.append(self)
P.Plugin.__init__ = init
caps = pkg.pipeline.Caps() if FB == 1 else pkg.pipeline.Caps.for_frames(FB)
w = pkg.synth.make_weights()
kw = dict(linear_compute=P.COMPUTE_SPLIT) if MODE == "split" else dict(linear_compute=P.COMPUTE_F16, head_dtype=torch.float16)
pipes, owned = [], []
for s in range(NS):
    k0 = len(ALL); pipes.append(pkg.pipeline.DsvtPipeline(w, caps=caps, device=dev, device_nms=True, frames=FB, **kw)); owned.append(ALL[k0:])
streams = [torch.cuda.Stream(dev) for _ in range(NS)]
ins = []
for s in range(NS):
    buf = np.zeros((1, FB * caps.N, 4), np.float32); ns = []
    for f in range(FB):
        p = pkg.synth.lidar_like(180000, RANK * 8 + FB * s + f); buf[0, f * caps.N:f * caps.N + len(p)] = p; ns.append(len(p))
    ins.append((torch.from_numpy(buf).to(dev), torch.tensor(ns, dtype=torch.int32, device=dev)))
def snap(s):
    d = {}
    for i, pl in enumerate(owned[s]):
        for j, (outs, ws) in enumerate(pl._cache.values()):
            for k, t in enumerate(outs):
                d[(i, pl.plugin_type, j, k)] = t.clone()
    d[("boxes",)] = pipes[s]._last[0].clone(); d[("cnt",)] = pipes[s]._last[1].clone()
    return d
refs, bad, first, detail = None, {}, {}, []
t_end = time.time() + float(os.environ.get("SECONDS", "40"))
it = 0
while time.time() < t_end:
    for s in range(NS):
        with torch.cuda.stream(streams[s]):
            pipes[s]._last = pipes[s].forward(*ins[s])
    torch.cuda.synchronize()
    cur = [snap(s) for s in range(NS)]
    if refs is None:
        refs = cur
    else:
        for s in range(NS):
            diffs = [k for k, v in cur[s].items() if not torch.equal(v, refs[s][k])]
            if diffs:
                bad.setdefault(s, []).append(it)
                first.setdefault((s, it), diffs[:6])
                if len(detail) < 4:
                    k = diffs[0]; a, b_ = cur[s][k].reshape(-1), refs[s][k].reshape(-1)
                    ne = (a != b_).nonzero().reshape(-1)
                    detail.append((s, it, k, int(ne.numel()), ne[:4].tolist(), a[ne[:4]].tolist(), b_[ne[:4]].tolist(), tuple(cur[s][k].shape)))
    it += 1
print("rank", RANK, "iters", it, "cnt", [r[("cnt",)].tolist() for r in refs], "iterations differing per stream:", {s: v[:10] for s, v in bad.items()} or "none")
for k, v in list(first.items())[:5]:
    print("   stream/iter", k, "first differing buffers", v)
for d in detail:
    print("   detail (stream, iter, buffer, differing elements, first indices, got, expected, shape):", d)
