"""In-kernel phase timestamps of the streamed encoder-MLP kernel (DSVT_MLP_TRACE=1 python tools/trace_mlp.py)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
pkg = G.load_package(); P = pkg.plugin
dev = torch.device("cuda:0")
MR, n, C = 65536, 34483, 192
w = pkg.synth.make_weights(with_bev=False)
lp = "module.backbone_3d.stage_0.2.encoder_list.0"
ln = lambda k: (w[lp + k + ".weight"], w[lp + k + ".bias"])
lns = [ln(".win_attn.norm1"), ln(".win_attn.norm2"), ln(".norm")]
mlp = P.add_encoder_mlp_op(w[lp + ".win_attn.self_attn.out_proj.weight"], w[lp + ".win_attn.self_attn.out_proj.bias"],
                           w[lp + ".win_attn.linear1.weight"], w[lp + ".win_attn.linear1.bias"],
                           w[lp + ".win_attn.linear2.weight"], w[lp + ".win_attn.linear2.bias"], lns, MR).set_zero_fill(False)
att = torch.randn((1, MR, C), device=dev).half(); x = torch.randn((1, MR, C), device=dev)
cnt = torch.tensor([n], dtype=torch.int32, device=dev)
for _ in range(3):
    mlp(att, cnt, x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    mlp(att, cnt, x)
e1.record(); torch.cuda.synchronize()
print(f"mlp {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch")
