"""Experiments on the dense BEV glue: MIOpen benchmark mode, fused conv+bias+relu ops."""
import os, sys, time
import numpy as np, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
G.build(); pkg = G.load_package()
dev = torch.device("cuda:0")
print("has miopen_convolution_relu:", hasattr(torch, "miopen_convolution_relu"), hasattr(torch.ops.aten, "miopen_convolution_relu"))
x = torch.randn(1, 128, 468, 468, device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
W = (torch.randn(128, 128, 3, 3, device=dev, dtype=torch.float16) * 0.03).contiguous(memory_format=torch.channels_last)
b = torch.randn(128, device=dev, dtype=torch.float16)
def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for bench in (False, True):
    torch.backends.cudnn.benchmark = bench
    print("benchmark", bench)
    print("  conv only        us", timeit(lambda: F.conv2d(x, W, None, 1, 1)))
    print("  conv+bias        us", timeit(lambda: F.conv2d(x, W, b, 1, 1)))
    print("  conv+bias+relu   us", timeit(lambda: F.relu(F.conv2d(x, W, b, 1, 1))))
    try:
        f = lambda: torch.miopen_convolution_relu(x, W, b, [1, 1], [1, 1], [1, 1], 1)
        print("  miopen_conv_relu us", timeit(f))
        d = (f() - F.relu(F.conv2d(x, W, b, 1, 1))).abs().max().item()
        print("   diff", d)
    except Exception as e:
        print("  miopen_convolution_relu failed:", repr(e)[:200])
    try:
        z = torch.randn_like(x)
        f2 = lambda: torch.miopen_convolution_add_relu(x, W, z, 1.0, b, [1, 1], [1, 1], [1, 1], 1)
        print("  miopen_conv_add_relu us", timeit(f2))
    except Exception as e:
        print("  miopen_convolution_add_relu failed:", repr(e)[:200])
