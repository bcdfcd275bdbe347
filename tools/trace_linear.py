"""In-kernel phase timestamps of the streamed fp16 linear kernel (DSVT_LINEAR_TRACE=1 python tools/trace_linear.py)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
pkg = G.load_package(); P = pkg.plugin
dev = torch.device("cuda:0")
MR, n = 65536, 34483
rng = np.random.default_rng(0)
cnt = torch.tensor([n], dtype=torch.int32, device=dev)
A = torch.randn((1, MR, 192), device=dev).half(); A2 = torch.randn((1, MR, 192), device=dev).half()
xy = torch.randn((1, MR, 2), device=dev)
W = (rng.standard_normal((576, 192)) / 14).astype(np.float32); b = np.zeros(576, np.float32)
qkv = P.add_linear_op(W, b, MR, add_cols=384, compute_type=1, input_half=True, output_mode=P.OUT_F16).set_zero_fill(False)
pe = P.add_linear_op(W[:192], b[:192], MR, compute_type=1, output_mode=P.OUT_F16, pe_weight=W[:192, :2].copy(), pe_bias=b[:192]).set_zero_fill(False)
for _ in range(3):
    qkv(A, cnt, A2); pe(xy, cnt)
torch.cuda.synchronize()
