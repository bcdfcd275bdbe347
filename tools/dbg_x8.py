import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); P = pkg.plugin
from tests.test_conv_mx_gpu import make_triple, x8_of
g = torch.Generator(device="cpu").manual_seed(5)
n, C, GX, GY = 700, 192, 40, 40
feat = torch.randn(1, 1000, C, generator=g) * torch.exp(torch.randn(1, 1000, 1, generator=g) * 2)
cells = torch.randperm(GX * GY, generator=g)[:1000]
coords = torch.zeros(1, 1000, 4, dtype=torch.int32)
coords[0, :, 2] = (cells // GX).int(); coords[0, :, 3] = (cells % GX).int()
cnt = torch.tensor([n], dtype=torch.int32)
bev = P.add_map_2_bev_op(1000, C, GX, GY, split_output=2)(feat.cuda(), coords.cuda(), cnt.cuda())[0].cpu()
want = torch.zeros(1, GY, GX, C)
want[0, coords[0, :n, 2].long(), coords[0, :n, 3].long()] = feat[0, :n]
t = make_triple(want)
print("hi/lo planes equal:", torch.equal(bev[..., :2*C].view(torch.int16), t[..., :2*C].view(torch.int16)))
a, b = x8_of(t, C), x8_of(bev, C)
bad = (a != b).nonzero()
print("x8 mismatches", bad.shape[0], "of", a.numel())
for i in bad[:12]:
    y, x, k = i[1].item(), i[2].item(), i[3].item()
    grp, hh, kind, j = k // 64, (k % 64) // 32, (k % 32) // 16, k % 16
    ch = grp * 32 + hh * 16 + j
    v = want[0, y, x, ch].item(); hi = float(np.float16(v)); 
    print(f"  cell ({y},{x}) byte {k} = ch {ch} kind {'hi8' if kind else 'lo8'}: v={v!r} hi={hi!r} (v-hi)*2048={(np.float32(v)-np.float32(hi))*2048!r} want 0x{a[tuple(i)].item():02x} have 0x{b[tuple(i)].item():02x}")
# ---- conv x8 output
import torch.nn.functional as F
from tests.test_conv_mx_gpu import nhwc
H, W, cin, cout = 52, 47, 128, 128
g = torch.Generator(device="cpu").manual_seed(7)
x = torch.randn(1, cin, H, W, generator=g) * 3.0
w = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(cin * 9)
op = P.add_conv2d_op(P.conv_weight_rows(w.numpy()), None, H, W, 3 * cin, cout, 3, 1, 1, split_output=2, split_input=2, out_channel_stride=3 * cout)
y3 = op(make_triple(nhwc(x)).cuda())[0].cpu()
v = y3[..., :cout].float() + y3[..., cout:2*cout].float()
t = make_triple(v)
a, b = x8_of(t, cout), x8_of(y3, cout)
af, bf = a.view(torch.float8_e4m3fn).float(), b.view(torch.float8_e4m3fn).float()
bad = ((af - bf).abs() > 0.126 * torch.maximum(af.abs(), bf.abs()) + 2.0**-9).nonzero()
print("conv x8 far mismatches", bad.shape[0], "of", a.numel(), "; any mismatch", (a != b).sum().item())
for i in bad[:12]:
    y, xx, k = i[1].item(), i[2].item(), i[3].item()
    grp, hh, kind, j = k // 64, (k % 64) // 32, (k % 32) // 16, k % 16
    ch = grp * 32 + hh * 16 + j
    hi = y3[0, y, xx, ch].item(); lo = y3[0, y, xx, cout + ch].item()
    print(f"  pix ({y},{xx}) byte {k} = ch {ch} kind {'hi8' if kind else 'lo8'}: hi={hi!r} lo={lo!r} lo*2048={lo*2048!r} want 0x{a[tuple(i)].item():02x} ({af[tuple(i)].item()}) have 0x{b[tuple(i)].item():02x} ({bf[tuple(i)].item()})")
