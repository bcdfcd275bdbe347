import os, sys
import numpy as np, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
G.build(); pkg = G.load_package(); P = pkg.plugin
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (H, cin, cout, k, s) in [(468, 128, 128, 3, 1), (468, 192, 128, 3, 1), (468, 384, 64, 3, 1), (468, 64, 320, 3, 1), (468, 320, 18, 3, 1),
                             (234, 128, 128, 3, 1), (117, 256, 256, 3, 1), (468, 192, 128, 1, 1)]:
    x = torch.randn(1, H, H, cin, device=dev, dtype=torch.float16)
    w = (torch.randn(cout, cin, k, k) / np.sqrt(cin * k * k)).numpy()
    op = P.add_conv2d_op(P.conv_weight_rows(w), np.zeros(cout, np.float32), H, H, cin, cout, k, s, k // 2, relu=True, out_f32=(cout == 18))
    us = timeit(lambda: op(x))
    xt = x.permute(0, 3, 1, 2); wt = torch.from_numpy(w).half().to(dev).contiguous(memory_format=torch.channels_last)
    us_t = timeit(lambda: F.relu(F.conv2d(xt, wt, None, s, k // 2)))
    fl = 2.0 * (H // s) ** 2 * cout * cin * k * k
    print(f"{H}^2 {cin}->{cout} k{k} s{s}: ours {us:7.1f} us {fl / us / 1e6:6.1f} TF | MIOpen+relu {us_t:7.1f} us {fl / us_t / 1e6:6.1f} TF")
