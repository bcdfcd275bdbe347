"""Per-stage GPU times of the frame pipeline (HIP events), for fp32 and fp16 dense head."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G

def main():
    G.build(); pkg = G.load_package()
    dev = torch.device("cuda:0")
    caps = pkg.pipeline.Caps()
    w = pkg.synth.make_weights()
    p = pkg.synth.lidar_like(180000, 0)
    buf = np.zeros((1, caps.N, 4), np.float32); buf[0, :p.shape[0]] = p
    pts = torch.from_numpy(buf).to(dev); n = torch.tensor([p.shape[0]], dtype=torch.int32, device=dev)
    ref = None
    from tools.vendor_dense import VendorDensePipeline
    for hd, lc, hh in [(torch.float32, 0, False), (torch.float16, 1, False), (torch.float16, 1, True), (torch.float32, 2, True)]:
        pipe = (pkg.pipeline.DsvtPipeline if hh else VendorDensePipeline)(w, caps=caps, device=dev, head_dtype=hd, linear_compute=lc)
        for _ in range(4):
            out = pipe.forward(pts, n)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        T = np.zeros(3)
        for it in range(10):
            ev[0].record(); st = pipe.voxel_stage(pts, n)
            ev[1].record(); x = pipe.backbone(st)
            ev[2].record(); boxes, cnt = pipe.head(x, st)
            ev[3].record(); torch.cuda.synchronize()
            T += [ev[i].elapsed_time(ev[i + 1]) for i in range(3)]
        T /= 10
        b = boxes[0].cpu().numpy().copy(); c = int(cnt[0])
        msg = ""
        if ref is None:
            ref = (b, c)
        else:
            sys.path.insert(0, ROOT)
            from tests.parity import match_boxes
            worst, un = match_boxes(b, c, ref[0], ref[1], tol=0.2)
            msg = f" vs fp32 head: count {c} vs {ref[1]}, max|diff| matched {worst:.3e}, unmatched {un}"
        print(f"head_dtype={hd} linear_compute={lc} hip_head={hh}: voxel_stage {T[0]:.3f} ms, backbone {T[1]:.3f} ms, head {T[2]:.3f} ms, total {T.sum():.3f} ms{msg}")

if __name__ == "__main__":
    main()
