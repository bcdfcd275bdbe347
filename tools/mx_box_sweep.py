"""Box error of the fp16 + fp8 head (head_mx) against the three-fp16-product head (within 1.5e-5 of the fp32 oracle) over many clouds:
FilterBoxByScore rows matched by class + nearest centre, max abs difference per column.  python tools/mx_box_sweep.py [seeds] [points]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
import bench
pkg = G.load_package(); P = pkg.plugin
dev = torch.device("cuda:0")
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 16
NP = int(sys.argv[2]) if len(sys.argv) > 2 else 180000
caps = pkg.pipeline.Caps()
w = pkg.synth.make_weights()
EXC = tuple(x for x in os.environ.get("EXCLUDE", "").split(",") if x)        # EXCLUDE=shared,heads0: those layers keep three fp16 products
# MLP_LO8=1 (ablate build: DSVT_HIP_LIB=dsvt-ai-trt_amd/libdsvt_hip_ablate.so): the encoder MLPs of the FIRST pipeline get w_lo rounded to e4m3 + a row scale
if os.environ.get("MLP_LO8"): os.environ["DSVT_MLP_LO8"] = "1"
pa = pkg.pipeline.DsvtPipeline(w, caps=caps, device=dev, linear_compute=P.COMPUTE_SPLIT, head_mx_exclude=EXC)                       # head_mx on (default)
os.environ.pop("DSVT_MLP_LO8", None)
pb = pkg.pipeline.DsvtPipeline(w, caps=caps, device=dev, linear_compute=P.COMPUTE_SPLIT, head_mx=False)
worst = {}
for s in range(NS):
    p = pkg.synth.lidar_like(NP, seed=s)
    buf = np.zeros((1, caps.N, 4), np.float32); buf[0, :len(p)] = p
    pts, n = torch.from_numpy(buf).to(dev), torch.tensor([len(p)], dtype=torch.int32, device=dev)
    ra, ca = pa.forward(pts, n); rb, cb = pb.forward(pts, n)
    torch.cuda.synchronize()
    e = bench.box_errors(ra[0].cpu().numpy(), int(ca[0]), rb[0].cpu().numpy(), int(cb[0]))
    print(f"seed {s}: " + " ".join(f"{k} {v:.2e}" for k, v in e.items() if k in ("xy", "z", "size", "yaw", "score")) + f" matched {e['matched']} boxes {e['boxes']}/{e['oracle_boxes']}", flush=True)
    for k in ("xy", "z", "size", "yaw", "score"):
        worst[k] = max(worst.get(k, 0.0), e[k])
print("worst over", NS, "clouds:", {k: float(f"{v:.3g}") for k, v in worst.items()})
