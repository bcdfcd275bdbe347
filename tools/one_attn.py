"""Run the fp16 set-attention plugin on the bench frame's set layout a few times (target of rocprofv3 --pmc passes)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
pkg = G.load_package(); P = pkg.plugin
dev = torch.device("cuda:0")
w = pkg.synth.make_weights(with_bev=False)
caps = pkg.pipeline.Caps()
pipe = pkg.pipeline.DsvtPipeline(w, caps=caps, with_head=False, linear_compute=P.COMPUTE_F16, head_dtype=torch.float16)
pts = pkg.synth.lidar_like(180000, 0)
buf = torch.zeros((1, caps.N, 4), device=dev); buf[0, :pts.shape[0]] = torch.from_numpy(pts).to(dev)
n = torch.tensor([pts.shape[0]], dtype=torch.int32, device=dev)
st = pipe.voxel_stage(buf, n)
inds, mask, S = st["gss"][0][0], st["gss"][0][1], st["gss"][0][2]
qkv = torch.randn((1, caps.P, 576), device=dev).half()
op = pipe.layers[(0, 0)]["attn"]
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    op(qkv, inds, mask, S)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    op(qkv, inds, mask, S)
e1.record(); torch.cuda.synchronize()
print(f"attention {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per launch (DSVT_ATTN_DBG={os.environ.get('DSVT_ATTN_DBG', '0')})")
