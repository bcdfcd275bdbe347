"""In-kernel slab stamps of conv_rows_kernel<SPL, TR> (ablation build):
    DSVT_HIP_LIB=dsvt-ai-trt_amd/libdsvt_hip_ablate.so DSVT_CONV_TRACE=1 python tools/trace_conv_rows.py [B] [H cin cout res]
per slab [start, MFMAs issued, own requests landed, barrier passed], per item [K loop done, epilogue done], shader cycles, waves 0 and 4 of four workgroups."""
import os, sys, subprocess
import numpy as np
if os.environ.get("TRACE_CHILD"):
    import torch
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    import __graft_entry__ as G
    pkg = G.load_package(); P = pkg.plugin
    dev = torch.device("cuda:0")
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    H, cin, cout, res = (int(v) for v in sys.argv[2:6]) if len(sys.argv) > 5 else (468, 128, 128, 0)
    rng = np.random.default_rng(0)
    w = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
    rows = P.split_weight_rows(P.conv_weight_rows(w), 9, cin)
    op = P.add_conv2d_op(rows, np.zeros(cout, np.float32), H, H, 3 * cin, cout, 3, 1, 1, relu=True, has_residual=bool(res), split_residual=bool(res),
                         split_output=4, split_input=1, out_channel_stride=3 * cout)
    x = (torch.randn(B, H, H, 3 * cin, device=dev) * 0.5).to(torch.float16); x[..., 2 * cin:] = x[..., :cin]
    x *= (torch.rand(B, H, H, 1, device=dev) < 0.35).to(torch.float16)
    r = (torch.randn(B, H, H, 3 * cout, device=dev) * 0.5).to(torch.float16)
    for _ in range(3):
        op(*([x, r] if res else [x]))
    torch.cuda.synchronize()
    sys.exit(0)
env = dict(os.environ, TRACE_CHILD="1")
out = subprocess.run([sys.executable, __file__] + sys.argv[1:], env=env, capture_output=True, text=True).stderr
lines = [l for l in out.splitlines() if l.startswith("[conv trace")]
lines = lines[-8:]                                  # the last launch's block
cin = int(sys.argv[3]) if len(sys.argv) > 5 else 128
NSLAB = 3 * (3 * cin // 32)
for l in lines:
    head, _, tail = l.partition("]")
    v = [int(x) for x in tail.split()]               # stamps after the kernel-start stamp: per item 4 x NSLAB slab stamps, [K loop done], [epilogue done]
    per = 4 * NSLAB + 2
    prev_end = 0
    for it in range(len(v) // per):
        seg = v[it * per:(it + 1) * per]
        a = np.array(seg[:4 * NSLAB]).reshape(NSLAB, 4)
        mf = a[:, 1] - a[:, 0]; wt = a[:, 2] - a[:, 1]; br = a[:, 3] - a[:, 2]
        epi = seg[-1] - seg[-2]
        print(f"{head}] item {it}: total {seg[-1] - prev_end:7d} cycles = MFMA blocks {mf.sum():6d} (median {int(np.median(mf))}, max {mf.max()}) + DMA waits {wt.sum():6d} (median {int(np.median(wt))}, max {wt.max()}) + "
              f"barriers {br.sum():6d} (median {int(np.median(br))}, max {br.max()}) + epilogue {epi} + rest {seg[-1] - prev_end - mf.sum() - wt.sum() - br.sum() - epi}")
        if it == 1:
            print("   per-slab (ky = s % 3) MFMA-block medians:", [int(np.median(mf[k::3])) for k in range(3)], "wait medians:", [int(np.median(wt[k::3])) for k in range(3)],
                  "barrier medians:", [int(np.median(br[k::3])) for k in range(3)])
        prev_end = seg[-1]
