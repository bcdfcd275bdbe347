"""One small frame through the HIP hot path on cuda:0, checked against the CPU oracle.
Called by __graft_entry__.smoke()."""
import numpy as np
import torch

from tests import cases
from tests.parity import match_boxes


def run(pkg, verbose=True):
    from oracle import oracle as O, dense_ref as D
    dev = torch.device("cuda:0")
    w = pkg.synth.make_weights()
    caps = pkg.pipeline.Caps.reference()
    pipe = pkg.pipeline.DsvtPipeline(w, caps=caps, device=dev, linear_compute=pkg.plugin.COMPUTE_SPLIT)      # the fp32-grade mode (bench.py's headline)
    pts, n = cases.load_frame("000000", caps.N)
    boxes, cnt = pipe.forward(torch.from_numpy(pts[None]).to(dev), torch.tensor([n], dtype=torch.int32, device=dev))
    torch.cuda.synchronize()
    boxes, cnt = boxes[0].cpu().numpy(), int(cnt[0])
    # integer path: bit-exact against the oracle
    st = pipe.voxel_stage(torch.from_numpy(pts[None]).to(dev), torch.tensor([n], dtype=torch.int32, device=dev))
    vox = O.points2features(pts, n, cases.p2f_cfg(cases.caps("ref")))
    assert int(st["P"][0]) == vox["P"] == 5504 and int(st["Nk"][0]) == vox["Nk"]
    assert np.array_equal(st["coords"][0].cpu().numpy().view(np.uint32), vox["coords"])
    assert int(st["gss"][0][2][0]) == 454          # src/dsvt-ai-trt.cpp:291 "1 454 36 192"
    # boxes: within 1e-3 of the fp32 oracle
    eb, ec = D.forward(pts, n, w, D.OracleCfg())
    worst, unmatched = match_boxes(boxes, cnt, eb, ec)
    if verbose:
        print(f"smoke: P={vox['P']} S12=454 boxes={cnt} (oracle {ec}) max|diff|={worst:.2e} unmatched={unmatched}")
    assert unmatched == 0 and worst < 1e-3, (worst, unmatched, cnt, ec)
