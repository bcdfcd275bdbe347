"""Repeatedly construct the split-precision pipeline in ONE process and hash its construction-time tables and the boxes of a golden frame
(optionally under tests/guard_alloc: DSVT_GUARD=1).  A sticky per-process difference of the boxes showed up ~1 run in 6 under the guard allocator."""
import os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
if os.environ.get("DSVT_GUARD", "0") != "0":
    alloc = torch.cuda.memory.CUDAPluggableAllocator(os.path.join(ROOT, "tests", "guard_alloc", "guard_alloc.so"), "guard_malloc", "guard_free")
    torch.cuda.memory.change_current_allocator(alloc)
import __graft_entry__ as G
import cases
pkg = G.load_package(); P = pkg.plugin
dev = torch.device("cuda:0")
w = pkg.synth.make_weights()
caps = pkg.pipeline.Caps()
crc = lambda t: f"{zlib.crc32(t.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes()):08x}"
pts, n = cases.load_frame("000000", caps.N)
pts_d, n_d = torch.from_numpy(pts[None]).to(dev), torch.tensor([n], dtype=torch.int32, device=dev)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for i in range(N):
    pipe = pkg.pipeline.DsvtPipeline(w, caps=caps, device=dev, device_nms=True, linear_compute=P.COMPUTE_SPLIT, head_mx=False)
    tabs = crc(torch.cat([pipe.pos_tables[k].reshape(-1) for k in sorted(pipe.pos_tables)]))
    st = pipe.voxel_stage(pts_d, n_d); torch.cuda.synchronize()
    v = crc(st["vfeat"]); 
    x = pipe.backbone(st); torch.cuda.synchronize()
    xb = crc(x)
    boxes, cnt = pipe.head(x, st); torch.cuda.synchronize()
    print(f"build {i}: tables {tabs} vfeat {v} backbone {xb} boxes {crc(boxes)} n {int(cnt[0])}", flush=True)
    del pipe, st, x, boxes
