"""A/B of conv_rows_kernel (round 6) against conv_wide_kernel<8, 8, 36, 4, 2, 2, SPL> on one three-product 3 x 3 layer, same box, alternating launches:
    python tools/ab_conv_rows.py [H] [cin] [cout] [images] [residual 0|1]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
pkg = G.load_package(); P = pkg.plugin
dev = torch.device("cuda:0")
H = int(sys.argv[1]) if len(sys.argv) > 1 else 468
cin = int(sys.argv[2]) if len(sys.argv) > 2 else 128
cout = int(sys.argv[3]) if len(sys.argv) > 3 else 128
B = int(sys.argv[4]) if len(sys.argv) > 4 else 4
res = bool(int(sys.argv[5])) if len(sys.argv) > 5 else False
g = torch.Generator(device="cpu").manual_seed(1)
w = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(cin * 9)
b = torch.randn(cout, generator=g) * 0.1
rows = P.split_weight_rows(P.conv_weight_rows(w.numpy()), 9, cin)
x = torch.relu(torch.randn(B, H, H, cin, generator=g) * 3.0)
hi = x.half(); lo = (x - hi.float()).half()
x3 = torch.cat([hi, lo, hi], -1).to(dev)
r3 = None
if res:
    r = torch.randn(B, H, H, cout, generator=g); rh = r.half()
    r3 = torch.cat([rh, (r - rh.float()).half(), rh], -1).to(dev)
ops, outs = [], []
for variant in (0, 1):
    ops.append(P.add_conv2d_op(rows, b.numpy(), H, H, 3 * cin, cout, 3, 1, 1, relu=True, has_residual=res, split_residual=1 if res else 0, split_input=1,
                               split_output=4, out_channel_stride=3 * cout, kernel_variant=variant))
    outs.append(torch.zeros((B, H, H, 3 * cout), dtype=torch.float16, device=dev))
args = [x3] + ([r3] if res else [])
for _ in range(3):
    for op, o in zip(ops, outs):
        op(*args, out=[o])
torch.cuda.synchronize()
print("bit-identical:", bool(torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))))
fl = 2.0 * B * H * H * cout * 9 * cin
for rep in range(3):
    for name, op, o in (("rows", ops[0], outs[0]), ("wide", ops[1], outs[1])):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        N = 20
        e0.record()
        for _ in range(N):
            op(*args, out=[o])
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / N
        print(f"{name}: {H}^2 {cin}->{cout} x{B} res={int(res)}  {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s of products")
