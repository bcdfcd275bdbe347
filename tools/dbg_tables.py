"""Which part of the split-mode position tables depends on what the allocator hands out: repeated construction under tests/guard_alloc
(DSVT_GUARD=1), dirty device memory recycled through hipMalloc (DIRTY=1), each table against a float64 product."""
import os, sys, zlib, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
if os.environ.get("DSVT_GUARD", "0") != "0":
    alloc = torch.cuda.memory.CUDAPluggableAllocator(os.path.join(ROOT, "tests", "guard_alloc", "guard_alloc.so"), "guard_malloc", "guard_free")
    torch.cuda.memory.change_current_allocator(alloc)
import __graft_entry__ as G
pkg = G.load_package(); P = pkg.plugin
from dsvt_ai_trt_amd.pipeline import fold_linear_bn, WINS
dev = torch.device("cuda:0")
w = pkg.synth.make_weights()
hip = ctypes.CDLL("libamdhip64.so")
def dirty(mb=256):
    ps = []
    for _ in range(8):
        p = ctypes.c_void_p(); assert hip.hipMalloc(ctypes.byref(p), mb << 20) == 0; hip.hipMemset(p, 0xFF, mb << 20); ps.append(p)
    hip.hipDeviceSynchronize()
    for p in ps: hip.hipFree(p)
ncell = 576
cnt = torch.tensor([ncell], dtype=torch.int32, device=dev)
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    if os.environ.get("DIRTY", "0") != "0": dirty()
    line = []
    for b in range(4):
        for l in range(2):
            pre = f"module.backbone_3d.input_layer.posembed_layers.0.{b}.{l}.position_embedding_head"
            Wa, ba = fold_linear_bn(w, pre + ".0", pre + ".1", 1e-5, bias=True)
            (wx, wy, _), _s = WINS[l]
            g = torch.zeros((1, ncell, 2), dtype=torch.float32)
            yy, xx = torch.meshgrid(torch.arange(wy), torch.arange(wx), indexing="ij")
            g[0, :wx * wy, 0] = xx.reshape(-1).float() - wx / 2; g[0, :wx * wy, 1] = yy.reshape(-1).float() - wy / 2
            gd = g.to(dev)
            op1 = P.add_linear_op(Wa, ba, ncell, activation=P.ACT_RELU); h1 = op1(gd, cnt)[0]
            op2 = P.add_linear_op(w[pre + ".3.weight"], w[pre + ".3.bias"], ncell); t = op2(h1, cnt)[0].clone()
            torch.cuda.synchronize()
            r1 = np.maximum(g[0].double().numpy() @ Wa.astype(np.float64).T + ba, 0)
            r2 = r1 @ w[pre + ".3.weight"].astype(np.float64).T + w[pre + ".3.bias"]
            e1 = np.abs(h1[0].cpu().double().numpy() - r1).max(); e2 = np.abs(t[0].cpu().double().numpy() - r2).max()
            line.append(f"{e1:.1e}/{e2:.1e}" + ("" if torch.equal(gd.cpu(), g) else "[H2D of g WRONG]") + ("" if int(cnt.cpu()[0]) == ncell else "[cnt WRONG]"))
    print("iter", it, " ".join(line), flush=True)
