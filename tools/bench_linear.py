"""Micro-benchmark of DsvtLinearPlugin shapes used by the pipeline (HIP events, 50 iterations each)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G

def main():
    G.build(); pkg = G.load_package(); P = pkg.plugin
    dev = torch.device("cuda:0")
    MR, n = 65536, 34483
    rng = np.random.default_rng(0)
    cnt = torch.tensor([n], dtype=torch.int32, device=dev)
    cases = [  # name, K, N, act, add_cols, n_ln
        ("posembed2 192->192", 192, 192, 0, 0, 0),
        ("qkv 192->576 +pos", 192, 576, 0, 384, 0),
        ("out 192->192 LN1", 192, 192, 0, 0, 1),
        ("fc1 192->384 gelu", 192, 384, 2, 0, 0),
        ("fc2 384->192 LN2", 384, 192, 0, 0, 2),
        ("fc2 384->192 LN3", 384, 192, 0, 0, 3),
    ]
    cts = [int(x) for x in (sys.argv[1:] or ["1"])]
    for name, K, N, act, add_cols, nln in cases:
        A = torch.randn((1, MR, K), device=dev); A2 = torch.randn((1, MR, K), device=dev)
        res = [torch.randn((1, MR, N), device=dev) for _ in range(nln)]
        W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32); b = rng.standard_normal(N).astype(np.float32) * 0.1
        lns = [(np.ones(N, np.float32), np.zeros(N, np.float32)) for _ in range(nln)]
        for ct in cts:
            op = P.add_linear_op(W, b, MR, activation=act, add_cols=add_cols, layer_norms=lns, compute_type=ct).set_zero_fill(False)
            args = [A, cnt] + ([A2] if add_cols else []) + res
            for _ in range(5): op(*args)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): op(*args)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 50 * 1e3
            flops = 2.0 * n * K * N
            byts = 4.0 * n * (K * (2 if add_cols else 1) + N * (1 + nln))
            print(f"{name:22s} ct={ct}: {us:7.1f} us  {flops / us / 1e6:7.1f} TFLOP/s  {byts / us / 1e3:7.1f} GB/s (algorithmic)")

if __name__ == "__main__":
    main()
