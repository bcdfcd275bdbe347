"""QKV (192 -> 576, x + position table on the first 384 columns) in split precision, rows of FR frames per launch (HIP events, 30 launches):
    python tools/bench_qkv_split.py [frames ...]     DSVT_HIP_LIB=<ablation build> DSVT_LINEAR_RESIDENT=0: the streamed kernel"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
pkg = G.load_package(); P = pkg.plugin
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
K, N = 192, 576
W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32); b = rng.standard_normal(N).astype(np.float32) * 0.1
for fr in [int(x) for x in (sys.argv[1:] or ["1", "4"])]:
    MR, n = 65536 * fr, 34362 * fr
    A = torch.randn((1, MR, K), device=dev); A2 = torch.randn((1, MR, K), device=dev)
    cnt = torch.tensor([n], dtype=torch.int32, device=dev)
    op = P.add_linear_op(W, b, MR, add_cols=384, compute_type=P.COMPUTE_SPLIT).set_zero_fill(False)
    for _ in range(5): op(A, cnt, A2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): out = op(A, cnt, A2)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    print(f"qkv split {fr} frames ({n} rows): {us:7.1f} us   {2.0 * n * K * N * 3 / us / 1e6:6.0f} TFLOP/s of fp16 MFMA issue   {4.0 * n * (2 * K + N) / us / 1e3:6.0f} GB/s algorithmic   checksum {float(out[0].double().sum()):.4f}")
