"""Split-precision encoder MLP, four frames per launch: the shipped <1, 8> kernel against the one-wave-per-SIMD experiments (ablation build:
DSVT_HIP_LIB=dsvt-ai-trt_amd/libdsvt_hip_ablate.so DSVT_MLP_SPLIT_VARIANT=<0|2|3|4> python tools/mlp_split_variants.py [frames]).  Prints the time per
launch and an FNV fingerprint of the output bits (all variants sum a row's products in the same order: the fingerprints must agree)."""
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
mlp = P.add_encoder_mlp_op(w[lp + ".win_attn.self_attn.out_proj.weight"], w[lp + ".win_attn.self_attn.out_proj.bias"],
                           w[lp + ".win_attn.linear1.weight"], w[lp + ".win_attn.linear1.bias"],
                           w[lp + ".win_attn.linear2.weight"], w[lp + ".win_attn.linear2.bias"], lns, MR, frames=FR, split_precision=True).set_zero_fill(False)
g = torch.Generator(device="cpu").manual_seed(5)
att = torch.randn((1, MR, C), generator=g).to(dev)
x = torch.randn((1, MR, C), generator=g).to(dev)
cnt = torch.tensor([n], dtype=torch.int32, device=dev)
for _ in range(3):
    out = mlp(att, cnt, x)
torch.cuda.synchronize()
o = out[0] if isinstance(out, (tuple, list)) else out
bits = o[0, :n].contiguous().view(torch.int32).cpu().numpy().astype(np.uint64)
fp = int((bits * np.arange(1, bits.size + 1, dtype=np.uint64).reshape(bits.shape)).sum() & np.uint64(0xFFFFFFFFFFFF))
if os.environ.get("MLP_SAVE"):
    np.save(os.environ["MLP_SAVE"], o[0, :n].cpu().numpy()[::7])
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for rep in range(3):
    e0.record()
    for _ in range(20):
        mlp(att, cnt, x)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 20 * 1e3)
print(f"variant {os.environ.get('DSVT_MLP_SPLIT_VARIANT', '0')}: {FR} frames, {n} rows: " + " / ".join(f"{t:.1f}" for t in ts) + f" us per launch, fingerprint {fp:012x}", flush=True)
