"""two pipelines on two streams, eager launches, same input every iteration: which stage's output ever differs from iteration 0?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
pkg = G.load_package(); P = pkg.plugin
FB = int(sys.argv[1]) if len(sys.argv) > 1 else 4
MODE = sys.argv[2] if len(sys.argv) > 2 else "f16"
NS = int(sys.argv[3]) if len(sys.argv) > 3 else 2
SEED0 = int(sys.argv[4]) if len(sys.argv) > 4 else 0
dev = torch.device("cuda:0")
caps = pkg.pipeline.Caps() if FB == 1 else pkg.pipeline.Caps.for_frames(FB)
w = pkg.synth.make_weights()
kw = dict(linear_compute=P.COMPUTE_SPLIT) if MODE == "split" else dict(linear_compute=P.COMPUTE_F16, head_dtype=torch.float16)
pipes = [pkg.pipeline.DsvtPipeline(w, caps=caps, device=dev, device_nms=True, frames=FB, **kw) for _ in range(NS)]
streams = [torch.cuda.Stream(dev) for _ in range(NS)]
ins = []
for s_ in range(NS):
    buf = np.zeros((1, FB * caps.N, 4), np.float32); ns = []
    for f in range(FB):
        p = pkg.synth.lidar_like(180000, SEED0 + FB * s_ + f); buf[0, f * caps.N:f * caps.N + len(p)] = p; ns.append(len(p))
    ins.append((torch.from_numpy(buf).to(dev), torch.tensor(ns, dtype=torch.int32, device=dev)))
def snap(s):
    pipe = pipes[s]
    pts, n = ins[s]
    st = pipe.voxel_stage(pts, n)
    out = dict(P=st["P"].clone(), Nk=st["Nk"].clone(), coords=st["coords"].clone(), pcnt=st["pcnt"].clone(), pidx=st["pidx"].clone(), feat=st["feat"].clone(), vfeat=st["vfeat"].clone())
    for k, g in enumerate(st["gss"]):
        out[f"inds{k}"] = g[0].clone(); out[f"mask{k}"] = g[1].clone(); out[f"S{k}"] = g[2].clone()
    x = pipe.backbone(st); out["x"] = x.clone()
    b, c = pipe.head(x, st); out["boxes"] = b.clone(); out["cnt"] = c.clone()
    return out
refs = []
for s in range(NS):          # each stream's reference: computed alone
    with torch.cuda.stream(streams[s]):
        refs.append(snap(s))
    torch.cuda.synchronize()
ref = refs[0]
bad = {}
for it in range(int(os.environ.get("ITERS", "40"))):
    outs = []
    for s in range(NS):
        with torch.cuda.stream(streams[s]):
            outs.append(snap(s))
    torch.cuda.synchronize()
    for s in range(NS):
        for k, v in outs[s].items():
            if not torch.equal(v, refs[s][k]):
                bad.setdefault(k, []).append((it, s))
import hashlib
print("sums", {k: hashlib.md5(v.cpu().numpy().tobytes()).hexdigest()[:8] for k, v in ref.items()}, "cnt", ref["cnt"].tolist())
print("mode", MODE, "frames", FB, "streams", NS, "differing tensors:", {k: v[:4] for k, v in bad.items()} if bad else "none")
