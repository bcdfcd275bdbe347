"""Per-layer durations of the 3 x 3 stride-1 layers of the dense stage in the three arithmetic variants (HIP events over REPS launches,
B images per launch):  f16 | split, three fp16 products over [hi | lo | hi] | split4, the same over [hi | lo | -] with the third plane aliased to plane 0
(round 5's default head wiring: split_input = 1, split_output = 4) | mx, fp16 + fp8 K loop over [hi | lo | x8].
    python tools/bench_conv_mx.py [B] [variants: f16,split,mx]      ONLY=<indices of LAYERS, comma separated>: those layers only (PMC passes)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
pkg = G.load_package(); P = pkg.plugin
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
variants = sys.argv[2].split(",") if len(sys.argv) > 2 else ["f16", "split", "mx"]
REPS = 6
OCC = float(os.environ.get("OCC", "0.35"))          # fraction of non-zero pixels (the BEV maps are sparse: all-random operands clock the chip down)
LAYERS = [  # H, cin, cout, residual, how many times in the network
    (468, 192, 128, False, 1), (468, 128, 128, True, 2), (468, 128, 128, False, 1), (234, 128, 128, True, 3), (234, 128, 128, False, 2),
    (117, 256, 256, True, 3), (117, 256, 256, False, 2), (468, 384, 64, False, 1), (468, 64, 320, False, 1)]
if os.environ.get("ONLY"):
    LAYERS = [LAYERS[int(i)] for i in os.environ["ONLY"].split(",")]
rng = np.random.default_rng(0)
tot = {v: 0.0 for v in variants}
for H, cin, cout, res, cnt in LAYERS:
    w = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
    b = np.zeros(cout, np.float32)
    line = f"{H}x{H} {cin:3d}->{cout:3d} res={int(res)} x{cnt}:"
    for v in variants:
        if v == "f16":
            op = P.add_conv2d_op(P.conv_weight_rows(w), b, H, H, cin, cout, 3, 1, 1, relu=True, has_residual=res)
            x = torch.randn(B, H, H, cin, device=dev, dtype=torch.float16)
            r = torch.randn(B, H, H, cout, device=dev, dtype=torch.float16)
        else:
            rows = P.conv_weight_rows(w) if v == "mx" else P.split_weight_rows(P.conv_weight_rows(w), 9, cin)
            op = P.add_conv2d_op(rows, b, H, H, 3 * cin, cout, 3, 1, 1, relu=True, has_residual=res, split_residual=res,
                                 split_output={"mx": 2, "split4": 4}.get(v, 1), split_input={"mx": 2, "split4": 1}.get(v, 0), out_channel_stride=3 * cout)
            x = (torch.randn(B, H, H, 3 * cin, device=dev) * 0.5).to(torch.float16)      # (any bit pattern is a valid operand; NaN-free fp8 bytes not required for timing)
            x[..., 2 * cin:] = torch.randint(0, 120, (B, H, H, cin), device=dev, dtype=torch.int16).view(torch.float16) if v == "mx" else x[..., :cin]
            r = (torch.randn(B, H, H, 3 * cout, device=dev) * 0.5).to(torch.float16)
        x *= (torch.rand(B, H, H, 1, device=dev) < OCC).to(torch.float16)
        args = [x, r] if res else [x]
        for _ in range(2):
            op(*args)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REPS):
            op(*args)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / REPS
        tot[v] += us * cnt
        line += f"  {v} {us:7.1f} us ({2 * B * H * H * 9 * cin * cout / us / 1e6:6.0f} TF)"
    print(line, flush=True)
print("dense-stage 3x3 stride-1 layers per forward of", B, "frames:", {v: round(t / 1e3, 3) for v, t in tot.items()}, "ms")
