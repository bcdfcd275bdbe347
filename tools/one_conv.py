"""Run one DsvtConv2dPlugin shape a few times (target of rocprofv3 --pmc passes).
    python tools/one_conv.py H cin cout k stride [reps]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
pkg = G.load_package(); P = pkg.plugin
H, cin, cout, k, s = [int(v) for v in sys.argv[1:6]]
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 5
dev = torch.device("cuda:0")
x = torch.randn(1, H, H, cin, device=dev, dtype=torch.float16)
w = (np.random.default_rng(0).standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
op = P.add_conv2d_op(P.conv_weight_rows(w), np.zeros(cout, np.float32), H, H, cin, cout, k, s, k // 2, relu=True)
for _ in range(reps):
    op(x)
torch.cuda.synchronize()
