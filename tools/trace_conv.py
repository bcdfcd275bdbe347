"""In-kernel slab timestamps of conv_wide_kernel (DSVT_CONV_TRACE=1 python tools/trace_conv.py [H cin cout]): per slab
[start, requests issued, MFMAs issued, own requests landed, barrier passed] in shader cycles, waves 0 and NW/2 of four workgroups."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
pkg = G.load_package(); P = pkg.plugin
dev = torch.device("cuda:0")
H, cin, cout = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (468, 128, 128)
x = torch.randn(1, H, H, cin, device=dev, dtype=torch.float16)
w = (torch.randn(cout, cin, 3, 3) / np.sqrt(cin * 9)).numpy()
op = P.add_conv2d_op(P.conv_weight_rows(w), np.zeros(cout, np.float32), H, H, cin, cout, 3, 1, 1, relu=True)
for _ in range(3):          # (every launch prints its stamps: read the last block of lines)
    op(x)
torch.cuda.synchronize()
