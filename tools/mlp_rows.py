"""Launch time of the encoder-MLP kernel against the row count (tile quantisation over the 256 CUs)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
pkg = G.load_package(); P = pkg.plugin
dev = torch.device("cuda:0")
MR, C = 65536, 192
w = pkg.synth.make_weights(with_bev=False)
lp = "module.backbone_3d.stage_0.2.encoder_list.0"
ln = lambda k: (w[lp + k + ".weight"], w[lp + k + ".bias"])
lns = [ln(".win_attn.norm1"), ln(".win_attn.norm2"), ln(".norm")]
mlp = P.add_encoder_mlp_op(w[lp + ".win_attn.self_attn.out_proj.weight"], w[lp + ".win_attn.self_attn.out_proj.bias"],
                           w[lp + ".win_attn.linear1.weight"], w[lp + ".win_attn.linear1.bias"],
                           w[lp + ".win_attn.linear2.weight"], w[lp + ".win_attn.linear2.bias"], lns, MR).set_zero_fill(False)
wi = w[lp + ".win_attn.self_attn.in_proj_weight"]; bi = w[lp + ".win_attn.self_attn.in_proj_bias"]
qkv = P.add_linear_op(wi, bi, MR, add_cols=2 * C, compute_type=P.COMPUTE_F16, input_half=True, output_mode=P.OUT_F16).set_zero_fill(False)
att = torch.randn((1, MR, C), device=dev).half(); x = torch.randn((1, MR, C), device=dev)
x16 = x.half(); pos16 = torch.randn((1, MR, C), device=dev).half()
for n in (16384, 24576, 32768, 33000, 34483, 36000, 40000, 49152, 65536):
    cnt = torch.tensor([n], dtype=torch.int32, device=dev)
    res = []
    for op, args in ((mlp, (att, cnt, x)), (qkv, (x16, cnt, pos16))):
        for _ in range(3):
            op(*args)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            op(*args)
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
    print(f"rows {n:6d} ({n / 128:6.1f} tiles)  mlp {res[0]:6.1f} us  qkv {res[1]:6.1f} us")
