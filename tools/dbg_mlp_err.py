"""max / mean error of the split-precision encoder MLP against the float64 restatement of the reference wiring (tests/test_split_kernels_gpu.py's case
(True, 65536, 34483)), for the library DSVT_HIP_LIB names: A/B of two builds at the level of rounding errors.  python tools/dbg_mlp_err.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
pkg = G.load_package(); P = pkg.plugin
from tests.test_split_kernels_gpu import _mlp_reference
from tests.test_plugins_gpu import dev, host, scalar
C, MR, n, b_ = 192, 65536, 34483, 1
rng = np.random.default_rng(7 * n + 1)
w = pkg.synth.make_weights(with_bev=False)
lp = f"module.backbone_3d.stage_0.{b_}.encoder_list.1"
ln = lambda k: (w[k + ".weight"], w[k + ".bias"])
lns = [ln(lp + ".win_attn.norm1"), ln(lp + ".win_attn.norm2"), ln(lp + ".norm"), ln(f"module.backbone_3d.residual_norm_stage_0.{b_}")]
att = np.zeros((MR, C), np.float32); att[:n] = rng.standard_normal((n, C))
x = np.zeros((MR, C), np.float32); x[:n] = rng.standard_normal((n, C))
xb = np.zeros((MR, C), np.float32); xb[:n] = rng.standard_normal((n, C))
for frames in (0, 1):
    mlp = P.add_encoder_mlp_op(w[lp + ".win_attn.self_attn.out_proj.weight"], w[lp + ".win_attn.self_attn.out_proj.bias"], w[lp + ".win_attn.linear1.weight"], w[lp + ".win_attn.linear1.bias"],
                               w[lp + ".win_attn.linear2.weight"], w[lp + ".win_attn.linear2.bias"], lns, MR, split_precision=True, frames=frames)
    got, = mlp(dev(att[None]), scalar(n), dev(x[None]), dev(xb[None]))
    torch.cuda.synchronize()
    g = host(got)[0]
    ref = _mlp_reference(att, x, xb, w, lp, b_, n, mimic_roundings=False, round_weights=False)
    err = np.abs(g[:n] - ref)
    print(f"frames={frames}: max {err.max():.3e} mean {err.mean():.3e} rows with err > 1e-6: {(err.max(1) > 1e-6).sum()} of {n}; fingerprint {float(np.abs(g[:n]).sum()):.6f}")
