"""The fp32-grade convolutions on the fp16 + fp8 K loop (round 4; csrc/conv.hip conv_wide_kernel<.., MX>, ConvArgs::x8_out / alias3).

A split tensor [hi | lo | x8] carries, in its third plane, the OCP e4m3 operands of the two correction terms of
a w = hi w_hi + lo w_hi + hi w_lo:  lo8 = e4m3(2^11 (a - hi)),  hi8 = e4m3(a),  per 32 channels 64 bytes
[lo8 0..15 | hi8 0..15 | lo8 16..31 | hi8 16..31].  The reference layers are convBnLELU / convBn + SUM + ReLU of
src/dsvt-ai-trt.cpp:149-246 in fp32; the tests compare with
  * a float64 convolution of the UNROUNDED operands (the bar the boxes need: fp16-operand error / ~10), and
  * a float64 EMULATION of the arithmetic (fp16 main product + the two fp8 products with the per-row weight exponent of
    DsvtConv2dPlugin::packMX), which a wrong tap pairing, byte order or scale would miss by ~2^-12 -- 50 x the bound used."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def e4m3(t):
    """decoded value of the OCP e4m3 byte the device writes: round to nearest even, saturated at +-448"""
    return t.float().clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float()


def e4m3_bytes(t):
    return t.float().clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)


def make_triple(x):
    """x [B,H,W,C] fp32 (CPU) -> fp16 [B,H,W,3C] = [hi | lo | x8]"""
    C = x.shape[-1]
    assert C % 32 == 0
    hi = x.half()
    d = x - hi.float()
    lo = d.half()
    lo8 = e4m3_bytes(d * 2048.0).reshape(*x.shape[:-1], C // 32, 2, 16)
    hi8 = e4m3_bytes(x).reshape(*x.shape[:-1], C // 32, 2, 16)
    x8 = torch.stack([lo8, hi8], dim=-2)                       # [.., group, half, kind, 16] -> bytes [lo 0..15 | hi 0..15 | lo 16..31 | hi 16..31]
    x8 = x8.reshape(*x.shape[:-1], 2 * C).contiguous().view(torch.float16)
    return torch.cat([hi, lo, x8], dim=-1).contiguous()


def split_value(t3, C):
    """[.., 3C] fp16 triple -> hi + lo in float64"""
    return t3[..., :C].double() + t3[..., C:2 * C].double()


def x8_of(t3, C):
    return t3[..., 2 * C:].contiguous().view(torch.uint8)


def assert_x8_plane(t3, C):
    """the x8 plane of a device-written triple against its own fp16 planes: lo8 = e4m3(2^11 lo), hi8 = e4m3(hi + lo); bytes equal up to the
    sign of zero, except where the fp16 rounding of lo moved the value across an e4m3 rounding boundary (rare, one code apart)"""
    lo, v = t3[..., C:2 * C].float(), t3[..., :C].float() + t3[..., C:2 * C].float()
    want = torch.stack([e4m3(lo * 2048.0).reshape(*lo.shape[:-1], C // 32, 2, 16), e4m3(v).reshape(*lo.shape[:-1], C // 32, 2, 16)], dim=-2)
    have = x8_of(t3, C).view(torch.float8_e4m3fn).float().reshape(want.shape)
    assert not torch.isnan(have).any()
    d = (want - have).abs()
    assert (d > 0.126 * torch.maximum(want.abs(), have.abs()) + 2.0 ** -9).sum().item() == 0
    assert (d != 0).float().mean().item() < 5e-3


def mx_emulation(x, w, b):
    """float64 value of what the kernel computes (before residual / ReLU): fp16 main product + the two fp8 correction products"""
    xh = x.half().float()
    a_lo8, a_hi8 = e4m3((x - xh) * 2048.0).double() / 2048.0, e4m3(x).double()
    wh = w.half().float()
    mx = w.abs().amax(dim=(1, 2, 3))
    e = torch.floor(torch.log2(torch.tensor(448.0) / mx)).clamp(-60, 60)
    s = torch.pow(torch.tensor(2.0), e).reshape(-1, 1, 1, 1)
    w_hi8 = e4m3(wh * s).double() / s.double()
    w_lo8 = e4m3((w - wh) * s * 2048.0).double() / (s.double() * 2048.0)
    y = F.conv2d(xh.double(), wh.double(), b.double(), 1, 1) + F.conv2d(a_lo8, w_hi8, None, 1, 1) + F.conv2d(a_hi8, w_lo8, None, 1, 1)
    return y


CASES = [
    # H, W, cin, cout, res, relu, images
    (52, 47, 128, 128, True, True, 1),        # basic-block conv2 + identity + ReLU: 8 rows x 64 channels, one row per wave (CT = 4, RW = 1)
    (52, 47, 192, 128, False, True, 1),       # first BEV conv
    (30, 33, 384, 64, False, True, 1),        # shared head conv (64 output channels: four channel tiles)
    (30, 33, 64, 320, False, True, 1),        # head stems (three 128-channel chunks, the last one partial)
    (37, 29, 256, 256, True, True, 2),        # third stage, two images
    (468, 468, 128, 128, True, True, 1),      # 16-row x 128-channel items (CT = 8): the 468 x 468 layers
    (234, 234, 128, 128, False, True, 1),     # 16-row x 64-channel items (CT = 4, RW = 2)
    (150, 140, 192, 128, False, False, 2),    # CT = 8 with six phases, two images
]


@pytest.mark.parametrize("H,W,cin,cout,res,relu,B", CASES)
def test_mx_conv_is_fp32_grade(pkg, H, W, cin, cout, res, relu, B):
    P = pkg.plugin
    g = torch.Generator(device="cpu").manual_seed(H * 1000 + cin + cout)
    x = torch.randn(B, cin, H, W, generator=g) * 3.0
    x = torch.relu(x) if (H + cin) % 2 else x                  # (the network's tensors are post-ReLU: half zeros)
    w = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(cin * 9)
    w = w * torch.exp(torch.randn(cout, 1, 1, 1, generator=g))   # rows of different magnitude: the per-row exponent matters
    b = torch.randn(cout, generator=g) * 0.1
    r = torch.randn(B, cout, H, W, generator=g) if res else None
    exact = F.conv2d(x.double(), w.double(), b.double(), 1, 1)
    emu = mx_emulation(x, w, b)
    if res:
        r3 = make_triple(nhwc(r))
        rv = split_value(r3, cout).permute(0, 3, 1, 2)
        exact, emu = exact + rv, emu + rv
    if relu:
        exact, emu = torch.relu(exact), torch.relu(emu)
    op = P.add_conv2d_op(P.conv_weight_rows(w.numpy()), b.numpy(), H, W, 3 * cin, cout, 3, 1, 1, relu=relu, has_residual=res, split_residual=res,
                         split_output=2, split_input=2, out_channel_stride=3 * cout)
    args = [make_triple(nhwc(x)).to(DEV)] + ([r3.to(DEV)] if res else [])
    y3 = op(*args)[0]
    torch.cuda.synchronize()
    y3 = y3.cpu()
    got = split_value(y3, cout).permute(0, 3, 1, 2)
    scale = exact.abs().max().item()
    e_emu, e_exact = (got - emu).abs().max().item() / scale, (got - exact).abs().max().item() / scale
    print(f"mx conv {H}x{W} {cin}->{cout}: vs emulation {e_emu:.2e}, vs exact fp64 {e_exact:.2e} of scale")
    assert e_emu < 8e-6, (e_emu, e_exact)          # fp32 accumulation order over 9 cin products + the fp8 blocks' internal sums
    assert e_exact < 1.5e-4, e_exact               # fp16 operands: ~1e-3; three fp16 products: ~1e-6
    # the x8 plane of the result: the encoder's bytes for the value the planes hold (v - hi is exact, so only a double rounding
    # through the fp16 lo plane can move a byte: rare, and then by one code)
    assert_x8_plane(y3, cout)
    again = op(*args)[0].cpu()
    assert torch.equal(again.view(torch.int16), y3.view(torch.int16))


@pytest.mark.parametrize("H,W,cin,cout,k,stride,res,relu,f32out", [
    (52, 47, 128, 256, 3, 2, False, True, False),     # strided (gather kernel)
    (52, 47, 128, 256, 1, 2, False, False, False),    # 1 x 1 downsample
    (52, 47, 192, 128, 1, 1, False, False, False),    # 1 x 1 shortcut of the first block (halo kernel)
    (25, 31, 320, 18, 3, 1, False, False, True),      # head outputs (narrow halo kernel, fp32 out)
    (52, 47, 128, 128, 3, 1, False, True, False),     # a 3 x 3 stride-1 layer kept on three fp16 products (conv_wide_kernel, 64-channel chunks)
    (150, 140, 64, 320, 3, 1, False, True, False),    # ... 128-channel chunks
])
def test_three_plane_kernels_read_plane_0_for_an_x8_third_plane(pkg, H, W, cin, cout, k, stride, res, relu, f32out):
    """split_input = 1: the kernels that still walk [hi | lo | hi] with [w_hi | w_hi | w_lo] rows get the same bits from a [hi | lo | x8]
    tensor (the third plane's phases read plane 0), and with split_output = 2 they write the x8 plane for their consumers."""
    P = pkg.plugin
    g = torch.Generator(device="cpu").manual_seed(H * 1000 + cin + cout + k)
    x = nhwc(torch.randn(1, cin, H, W, generator=g) * 3.0)
    w = torch.randn(cout, cin, k, k, generator=g) / np.sqrt(cin * k * k)
    b = torch.randn(cout, generator=g) * 0.1
    rows = P.split_weight_rows(P.conv_weight_rows(w.numpy()), k * k, cin)
    t_mx = make_triple(x)
    t_old = torch.cat([t_mx[..., :2 * cin], t_mx[..., :cin]], dim=-1).contiguous()
    kw = dict(relu=relu, out_f32=f32out, out_channel_stride=cout if f32out else 3 * cout)
    old = P.add_conv2d_op(rows, b.numpy(), H, W, 3 * cin, cout, k, stride, k // 2, split_output=0 if f32out else 1, **kw)(t_old.to(DEV))[0].cpu()
    new = P.add_conv2d_op(rows, b.numpy(), H, W, 3 * cin, cout, k, stride, k // 2, split_output=0 if f32out else 2, split_input=1, **kw)(t_mx.to(DEV))[0].cpu()
    if f32out:
        # (split_input = 1 puts the block-diagonal head outputs on conv3x3_grouped_narrow_split_kernel, which sums (hi w_hi, hi w_lo, lo w_hi) per
        # output where the dense walk of `old` sums (hi w_hi, lo w_hi, hi w_lo): the same products in another fp32 order)
        assert (old - new).abs().max().item() <= 2e-6 * old.abs().max().item()
        return
    if k == 1 and stride == 1:
        # (round 5: split_input = 1 puts the 1 x 1 stride-1 layers on conv1x1_resident_split_kernel, which sums per k-step (hi w_hi, lo w_hi, hi w_lo) where the
        # halo kernel of `old` sums plane by plane: the same products in another fp32 order)
        vo = old[..., :cout].double() + old[..., cout:2 * cout].double(); vn = new[..., :cout].double() + new[..., cout:2 * cout].double()
        assert (vo - vn).abs().max().item() <= 2e-6 * vo.abs().max().item()
    else:
        assert torch.equal(old[..., :2 * cout].view(torch.int16), new[..., :2 * cout].view(torch.int16))
    assert_x8_plane(new, cout)


def test_map2bev_writes_the_x8_plane(pkg):
    P = pkg.plugin
    g = torch.Generator(device="cpu").manual_seed(5)
    n, C, GX, GY = 700, 192, 40, 40
    feat = torch.randn(1, 1000, C, generator=g) * torch.exp(torch.randn(1, 1000, 1, generator=g) * 2)
    cells = torch.randperm(GX * GY, generator=g)[:1000]
    coords = torch.zeros(1, 1000, 4, dtype=torch.int32)
    coords[0, :, 2] = (cells // GX).int(); coords[0, :, 3] = (cells % GX).int()
    cnt = torch.tensor([n], dtype=torch.int32)
    bev = P.add_map_2_bev_op(1000, C, GX, GY, split_output=2)(feat.to(DEV), coords.to(DEV), cnt.to(DEV))[0].cpu()
    want = torch.zeros(1, GY, GX, C)
    want[0, coords[0, :n, 2].long(), coords[0, :n, 3].long()] = feat[0, :n]
    assert torch.equal(bev.view(torch.int16), make_triple(want).view(torch.int16))
    # split_output = 3 (round 5, the default head's map): [hi | lo | -], the third plane stays as the fill left it (zeros)
    bev3 = P.add_map_2_bev_op(1000, C, GX, GY, split_output=3)(feat.to(DEV), coords.to(DEV), cnt.to(DEV))[0].cpu()
    assert torch.equal(bev3[..., :2 * C].view(torch.int16), bev[..., :2 * C].view(torch.int16)) and not bev3[..., 2 * C:].any()


@pytest.mark.parametrize("H,W,B", [(468, 468, 1), (61, 45, 3), (7, 5, 2)])
def test_block_diagonal_head_outputs_at_fp32_grade_on_the_grouped_split_kernel(pkg, H, W, B):
    """The CenterHead's five output convolutions as ONE 320 -> 18 layer over a split input [hi | lo | x8] (split_input = 1, rows [w_hi | w_hi | w_lo]):
    conv3x3_grouped_narrow_split_kernel (a head's w_hi and w_lo tiles resident, its hi and lo halos streamed) against the float64 convolution of the fp32
    operands (fp32 grade: 2e-6 of scale, where one fp16 product sits at 1e-3) and against the dense three-plane walk, reached by a structural zero that is
    non-zero in fp32 and zero in both fp16 parts (same products, another summation order: 2e-6)."""
    P = pkg.plugin
    g = torch.Generator(device="cpu").manual_seed(H * 11 + W)
    heads = [2, 1, 3, 2, 10]
    cin, cout = 64 * len(heads), sum(heads)
    w = torch.zeros(cout, cin, 3, 3)
    n0 = 0
    for h, n in enumerate(heads):
        w[n0:n0 + n, 64 * h:64 * (h + 1)] = torch.randn(n, 64, 3, 3, generator=g) / 24.0
        n0 += n
    b = torch.randn(cout, generator=g) * 0.1
    x = nhwc(torch.randn(B, cin, H, W, generator=g) * 2.0)
    t_mx = make_triple(x).to(DEV)
    mk = lambda ww: P.add_conv2d_op(P.split_weight_rows(P.conv_weight_rows(ww.numpy()), 9, cin), b.numpy(), H, W, 3 * cin, cout, 3, 1, 1, out_f32=True,
                                    out_channel_stride=cout, split_input=1)
    got = mk(w)(t_mx)[0].cpu()
    w_dense = w.clone(); w_dense[0, 100, 1, 1] = 1e-30            # channel 0 now "reads" group 1 as well: the dense kernel; hi = lo = 0 in fp16
    ref_dense = mk(w_dense)(t_mx)[0].cpu()
    assert got.dtype == torch.float32 and tuple(got.shape) == (B, H, W, cout)
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=1).permute(0, 2, 3, 1)
    scale = ref.abs().max().item()
    assert (got.double() - ref).abs().max().item() < 2e-6 * scale
    assert (got - ref_dense).abs().max().item() < 2e-6 * scale


def test_split_output_4_writes_hi_and_lo_and_leaves_the_third_plane_alone(pkg):
    """split_output = 4 (a tensor that is only ever a residual): the hi and lo planes are those of split_output = 2, the third plane keeps what the buffer held
    (1 x 1 layer on the fp16 + fp8 loop, the first block's shortcut; and a 3 x 3 layer on three fp16 products)."""
    P = pkg.plugin
    g = torch.Generator(device="cpu").manual_seed(41)
    for k, cin, cout, si in ((1, 192, 128, 2), (3, 64, 128, 1)):
        H, W = 37, 45
        x = nhwc(torch.randn(1, cin, H, W, generator=g) * 2.0)
        w = torch.randn(cout, cin, k, k, generator=g) / np.sqrt(cin * k * k); b = torch.randn(cout, generator=g) * 0.1
        rows = P.conv_weight_rows(w.numpy()) if si == 2 else P.split_weight_rows(P.conv_weight_rows(w.numpy()), k * k, cin)
        t = make_triple(x).to(DEV)
        mk = lambda so: P.add_conv2d_op(rows, b.numpy(), H, W, 3 * cin, cout, k, 1, k // 2, split_output=so, split_input=si, out_channel_stride=3 * cout)
        full = mk(2)(t)[0].cpu()
        buf = torch.full((1, H, W, 3 * cout), 7.0, dtype=torch.float16, device=DEV)
        got = mk(4)(t, out=[buf])[0].cpu()
        assert torch.equal(got[..., :2 * cout].view(torch.int16), full[..., :2 * cout].view(torch.int16))
        assert (got[..., 2 * cout:] == 7.0).all()


def test_mx_plugin_refuses_what_the_kernel_does_not_serve(pkg):
    P = pkg.plugin
    w = np.zeros((128, 9 * 128), np.float32)
    for kw in (dict(kernel_size=1, padding=0, stride=2), dict(stride=2), dict(kernel_size=1, padding=1)):
        args = dict(kernel_size=3, stride=1, padding=1)
        args.update(kw)
        with pytest.raises(Exception):
            P.add_conv2d_op(w[:, :(args["kernel_size"] ** 2) * 128], None, 20, 20, 384, 128, split_input=2, split_output=2, out_channel_stride=384, **args)
    with pytest.raises(Exception):      # a 1 x 1 layer whose weight rows are not whole 128-row chunks
        P.add_conv2d_op(w[:64, :128], None, 20, 20, 384, 64, 1, 1, 0, split_input=2, split_output=2, out_channel_stride=192)
    with pytest.raises(Exception):      # 32 output channels: the narrow halo kernel's layer
        P.add_conv2d_op(w[:32], None, 20, 20, 384, 32, 3, 1, 1, split_input=2, split_output=2, out_channel_stride=96)


def test_mx_conv_blob_round_trip(pkg):
    """serialize -> deserialize keeps split_input / split_output (four-int trailer ending in the magic word); a padded blob is refused"""
    P = pkg.plugin
    g = torch.Generator(device="cpu").manual_seed(3)
    w = torch.randn(64, 64, 3, 3, generator=g) / 24
    op = P.add_conv2d_op(P.conv_weight_rows(w.numpy()), None, 9, 11, 192, 64, 3, 1, 1, split_input=2, split_output=2, out_channel_stride=192)
    x3 = make_triple(nhwc(torch.randn(1, 64, 9, 11, generator=g))).to(DEV)
    y = op(x3)[0].cpu()
    blob = op.serialize()
    op2 = P.Plugin.deserialize("DsvtConv2dPlugin", blob)
    assert torch.equal(op2(x3)[0].cpu().view(torch.int16), y.view(torch.int16))
    with pytest.raises(Exception):
        P.Plugin.deserialize("DsvtConv2dPlugin", blob + b"\0\0\0\0")


def test_split_output_3_leaves_the_lo_plane_alone(pkg):
    """[hi | - | x8]: a tensor only the fp16 + fp8 K loop reads -- the hi and x8 planes are split_output = 2's, the lo plane keeps whatever it held"""
    P = pkg.plugin
    g = torch.Generator(device="cpu").manual_seed(9)
    H, W, cin, cout = 21, 37, 64, 128
    w = torch.randn(cout, cin, 3, 3, generator=g) / 24
    x3 = make_triple(nhwc(torch.randn(2, cin, H, W, generator=g))).to(DEV)
    kw = dict(split_input=2, out_channel_stride=3 * cout, relu=True)
    full = P.add_conv2d_op(P.conv_weight_rows(w.numpy()), None, H, W, 3 * cin, cout, 3, 1, 1, split_output=2, **kw)(x3)[0].cpu()
    out = torch.full((2, H, W, 3 * cout), 7.0, dtype=torch.float16, device=DEV)
    P.add_conv2d_op(P.conv_weight_rows(w.numpy()), None, H, W, 3 * cin, cout, 3, 1, 1, split_output=3, **kw)(x3, out=[out])
    torch.cuda.synchronize()
    out = out.cpu()
    assert torch.equal(out[..., :cout].view(torch.int16), full[..., :cout].view(torch.int16))
    assert torch.equal(out[..., 2 * cout:].view(torch.int16), full[..., 2 * cout:].view(torch.int16))
    assert (out[..., cout:2 * cout] == 7.0).all()


def mx_rows_emulation(x, rows, bias_rows):
    """x [N, C] fp32, rows [R, C] fp32 -> float64 [N, R]: fp16 main product + the two e4m3 correction products, one exponent per weight row"""
    xh = x.half().float()
    a_lo8, a_hi8 = e4m3((x - xh) * 2048.0).double() / 2048.0, e4m3(x).double()
    wh = rows.half().float()
    e = torch.floor(torch.log2(torch.tensor(448.0) / rows.abs().amax(dim=1))).clamp(-60, 60)
    sc = torch.pow(torch.tensor(2.0), e).reshape(-1, 1)
    w_hi8 = e4m3(wh * sc).double() / sc.double()
    w_lo8 = e4m3((rows - wh) * sc * 2048.0).double() / (sc.double() * 2048.0)
    return xh.double() @ wh.double().T + a_lo8 @ w_hi8.T + a_hi8 @ w_lo8.T + bias_rows.double()


@pytest.mark.parametrize("H,W,cin,cout,up,B", [
    (52, 47, 192, 128, 1, 1),       # the 1 x 1 shortcut of the first block
    (26, 23, 128, 128, 2, 1),       # second deblock: ConvTranspose 2 x 2 / 2 as a 1 x 1 layer with a pixel shuffle
    (13, 11, 256, 128, 4, 2),       # third deblock, two images
    (468, 468, 128, 128, 1, 1),     # first deblock at full size
    (40, 37, 320, 128, 1, 1),       # channel counts conv1x1_resident_mx_kernel does not take (it serves C = 128 / 192 / 256): conv_halo_kernel<.., MX>
    (33, 30, 64, 128, 2, 1),
])
def test_mx_1x1_layers_and_deblocks(pkg, H, W, cin, cout, up, B):
    """1 x 1 layers of the fp32-grade dense stage on the fp16 + fp8 K loop (split_input = 2: conv1x1_resident_mx_kernel in one, three and two passes for C = 128 /
    192 / 256, conv_halo_kernel<.., MX> for other widths), deconvBnLELU included (stride == kernel ConvTranspose,
    src/dsvt-ai-trt.cpp:217-246), written into a channel slice of the concat buffer with split_output = 3 ([hi | - | x8])"""
    P = pkg.plugin
    g = torch.Generator(device="cpu").manual_seed(H * 100 + cin + up)
    x = torch.relu(torch.randn(B, H, W, cin, generator=g) * 3.0)
    w = torch.randn(cin, cout, up, up, generator=g) / np.sqrt(cin) * torch.exp(torch.randn(1, cout, 1, 1, generator=g))
    b = torch.randn(cout, generator=g) * 0.1
    rows = torch.from_numpy(P.deconv_weight_rows(w.numpy()))                     # [(dy, dx, co)][cin]
    brow = b.repeat(up * up)
    xf = x.reshape(-1, cin)
    emu = torch.relu(mx_rows_emulation(xf, rows, brow))
    exact = torch.relu(xf.double() @ rows.double().T + brow.double())
    shuffle = lambda y: y.reshape(B, H, W, up, up, cout).permute(0, 1, 3, 2, 4, 5).reshape(B, H * up, W * up, cout)
    emu, exact = shuffle(emu), shuffle(exact)
    plane, off = 384, 128
    op = P.add_conv2d_op(rows.numpy(), b.numpy(), H, W, 3 * cin, cout, 1, 1, 0, pixel_shuffle=up, relu=True, split_input=2, split_output=3,
                         out_channel_stride=3 * plane, out_channel_offset=off)
    cat = torch.full((B, H * up, W * up, 3 * plane), 7.0, dtype=torch.float16, device=DEV)
    op(make_triple(x).to(DEV), out=[cat])
    torch.cuda.synchronize()
    cat = cat.cpu()
    got = cat[..., off:off + cout].double()
    lo8 = cat[..., 2 * plane:].contiguous().view(torch.uint8).reshape(B, H * up, W * up, plane // 32, 2, 2, 16)[..., 0, :]      # lo8 bytes [.., group, half, 16]
    got = got + lo8.reshape(B, H * up, W * up, plane).view(torch.float8_e4m3fn).float()[..., off:off + cout].double() / 2048.0     # hi + the x8 plane's lo part (4 bits)
    scale = exact.abs().max().item()
    e_emu, e_exact = (got - emu).abs().max().item() / scale, (got - exact).abs().max().item() / scale
    print(f"mx 1x1 {H}x{W} {cin}->{cout} up{up}: vs emulation {e_emu:.2e}, vs exact fp64 {e_exact:.2e} of scale")
    assert e_emu < 4e-5 and e_exact < 1.5e-4, (e_emu, e_exact)        # (hi + lo8 / 2^11 carries the value to 2^-16: the lo plane is not written)
    untouched = torch.ones(3 * plane, dtype=torch.bool)
    untouched[off:off + cout] = False; untouched[2 * plane + off:2 * plane + off + cout] = False
    assert (cat[..., untouched] == 7.0).all()


@pytest.mark.parametrize("H,W,c,B", [(468, 468, 128, 1), (61, 45, 256, 2), (30, 33, 64, 1)])
def test_residual_without_a_lo_plane_comes_from_the_x8_plane(pkg, H, W, c, B):
    """split_residual = 2: the residual tensor is [hi | - | x8] (written with split_output = 3); its value is hi + 2^-11 lo8, i.e. the fp32 value
    to 2^-15 relative.  The lo plane of the tensor handed in holds garbage that must not be read."""
    P = pkg.plugin
    g = torch.Generator(device="cpu").manual_seed(H + c)
    x = torch.relu(torch.randn(B, c, H, W, generator=g) * 3.0)
    w = torch.randn(c, c, 3, 3, generator=g) / np.sqrt(c * 9)
    r = torch.randn(B, c, H, W, generator=g) * 4.0
    r3 = make_triple(nhwc(r))
    lo8 = x8_of(r3, c).reshape(B, H, W, c // 32, 2, 2, 16)[..., 0, :].reshape(B, H, W, c).view(torch.float8_e4m3fn).float()
    rv = (r3[..., :c].double() + lo8.double() / 2048.0).permute(0, 3, 1, 2)
    emu = torch.relu(mx_emulation(x, w, torch.zeros(c)) + rv)
    r3[..., c:2 * c] = 123.0                                                       # the lo plane nobody may read
    op = P.add_conv2d_op(P.conv_weight_rows(w.numpy()), None, H, W, 3 * c, c, 3, 1, 1, relu=True, has_residual=True, split_residual=2,
                         split_output=2, split_input=2, out_channel_stride=3 * c)
    y3 = op(make_triple(nhwc(x)).to(DEV), r3.to(DEV))[0].cpu()
    got = split_value(y3, c).permute(0, 3, 1, 2)
    err = (got - emu).abs().max().item() / emu.abs().max().item()
    assert err < 8e-6, err
    assert ((rv - r.double()).abs() <= 2.0 ** -15 * r.double().abs() + 2.0 ** -20).all()      # what the x8 residual loses against the fp32 one: 2^-4 of lo <= 2^-11 |v|
