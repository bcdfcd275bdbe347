import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as G
pkg = G.load_package(); P = pkg.plugin
dev = torch.device("cuda:0")
MR, n, C = 65536, 34483, 192
rng = np.random.default_rng(0)
xy = [torch.randn((1, MR, 2), device=dev) for _ in range(2)]
cnt = torch.tensor([n], dtype=torch.int32, device=dev)
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for L in (1, 2, 4, 8):
    Wa = [(rng.standard_normal((C, 2)) * 0.5).astype(np.float32) for _ in range(L)]
    ba = [np.zeros(C, np.float32) for _ in range(L)]
    Wb = [(rng.standard_normal((C, C)) / 14).astype(np.float32) for _ in range(L)]
    bb = [np.zeros(C, np.float32) for _ in range(L)]
    op = P.add_pos_embed_op(MR, [l % 2 for l in range(L)], Wa, ba, Wb, bb).set_zero_fill(False)
    print("L", L, "batched %.1f us" % t(lambda: op(cnt, *xy)))
one = P.add_linear_op(Wb[0], bb[0], MR, compute_type=P.COMPUTE_F16, output_mode=P.OUT_F16, pe_weight=Wa[0], pe_bias=ba[0]).set_zero_fill(False)
print("single launch %.1f us" % t(lambda: one(xy[0], cnt)))
