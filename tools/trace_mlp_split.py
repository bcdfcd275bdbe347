"""Ablations of the split-precision encoder MLP at four frames per launch (ablation build: DSVT_HIP_LIB=dsvt-ai-trt_amd/libdsvt_hip_ablate.so
DSVT_MLP_DBG=<bits> python tools/trace_mlp_split.py): 1 no LN1, 2 no GELU, 4 no final LNs, 8 no stores, 16 no MFMA."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
pkg = G.load_package(); P = pkg.plugin
dev = torch.device("cuda:0")
FR = int(sys.argv[1]) if len(sys.argv) > 1 else 4
MR, n, C = 65536 * FR, 34362 * FR, 192
w = pkg.synth.make_weights(with_bev=False)
lp = "module.backbone_3d.stage_0.2.encoder_list.0"
ln = lambda k: (w[lp + k + ".weight"], w[lp + k + ".bias"])
lns = [ln(".win_attn.norm1"), ln(".win_attn.norm2"), ln(".norm")]
for split in (True, False):
    mlp = P.add_encoder_mlp_op(w[lp + ".win_attn.self_attn.out_proj.weight"], w[lp + ".win_attn.self_attn.out_proj.bias"],
                               w[lp + ".win_attn.linear1.weight"], w[lp + ".win_attn.linear1.bias"],
                               w[lp + ".win_attn.linear2.weight"], w[lp + ".win_attn.linear2.bias"], lns, MR, frames=FR, split_precision=split).set_zero_fill(False)
    att = torch.randn((1, MR, C), device=dev); att = att if split else att.half()
    x = torch.randn((1, MR, C), device=dev)
    cnt = torch.tensor([n], dtype=torch.int32, device=dev)
    for _ in range(3):
        mlp(att, cnt, x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        mlp(att, cnt, x)
    e1.record(); torch.cuda.synchronize()
    print(f"{'split' if split else 'f16  '} mlp, {FR} frames: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch (DSVT_MLP_DBG={os.environ.get('DSVT_MLP_DBG', '0')})", flush=True)
