"""DsvtConv2dPlugin (NHWC fp16 implicit-GEMM convolution) against PyTorch's convolution on the same
fp16-rounded operands.  fp32 accumulation on both sides: differences are summation order + the final
fp16 rounding of the output (2^-11 relative)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ref(x, w, b, stride, pad, res=None, relu=False):
    """the reference convolution on the CPU (PyTorch fp32, the arithmetic oracle/dense_ref.py restates the TensorRT layers with); the result
    goes back to the device only to be subtracted there.  (Until round 4 this ran F.conv2d on the GPU, i.e. MIOpen checked our kernels.)"""
    dev = x.device
    y = F.conv2d(x.float().cpu(), w.float().cpu(), None if b is None else b.float().cpu(), stride, pad)
    if res is not None:
        y = y + res.float().cpu()
    return (torch.relu(y) if relu else y).to(dev)


def nhwc(t):      # NCHW tensor -> contiguous [1,H,W,C]
    return t.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("H,W,cin,cout,k,stride,res,relu", [
    (52, 47, 128, 128, 3, 1, True, True),      # basic block conv2 + identity + ReLU (KC=128)
    (52, 47, 192, 128, 3, 1, False, True),     # first BEV conv (KC=96)
    (52, 47, 128, 256, 3, 2, False, True),     # strided, two output-channel chunks
    (52, 47, 128, 256, 1, 2, False, False),    # 1x1 downsample
    (30, 33, 384, 64, 3, 1, False, True),      # shared head conv
    (30, 33, 64, 320, 3, 1, False, True),      # five head stems at once (KC=64, 3 chunks, last one partial)
    # 1 x 1 layers on the resident-weights streaming kernel (conv1x1_resident_kernel): fewer pixels than one 16-pixel tile, odd sizes,
    # six k-steps (192 input channels), stride 2 with an odd input size, 256 output channels
    (5, 3, 128, 128, 1, 1, False, True),
    (7, 9, 192, 128, 1, 1, False, True),
    (37, 41, 256, 128, 1, 1, False, False),
    (7, 9, 128, 128, 1, 2, False, False),
    (117, 117, 128, 256, 1, 2, False, True),
])
def test_conv_matches_torch(pkg, H, W, cin, cout, k, stride, res, relu):
    P = pkg.plugin
    g = torch.Generator(device="cpu").manual_seed(H * 1000 + cin + cout)
    x = (torch.randn(1, cin, H, W, generator=g)).half().to(DEV)
    w = (torch.randn(cout, cin, k, k, generator=g) / np.sqrt(cin * k * k)).half().to(DEV)
    b = (torch.randn(cout, generator=g) * 0.1).to(DEV)
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    r = torch.randn(1, cout, Ho, Wo, generator=g).half().to(DEV) if res else None
    ref = _ref(x, w, b, stride, pad, r, relu)
    op = P.add_conv2d_op(P.conv_weight_rows(w.float().cpu().numpy()), b.cpu().numpy(), H, W, cin, cout, k, stride, pad,
                         relu=relu, has_residual=res)
    args = [nhwc(x)] + ([nhwc(r)] if res else [])
    o = op(*args)[0]
    torch.cuda.synchronize()
    got = o.permute(0, 3, 1, 2).float()
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() < 2e-3 * scale


def test_conv_f32_output_partial_channels(pkg):
    """heads' last conv: 320 -> 18 channels (not a multiple of 4/16), fp32 output."""
    P = pkg.plugin
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn(1, 320, 25, 31, generator=g).half().to(DEV)
    w = (torch.randn(18, 320, 3, 3, generator=g) / 50).half().to(DEV)
    b = torch.randn(18, generator=g).to(DEV)
    ref = _ref(x, w, b, 1, 1)
    op = P.add_conv2d_op(P.conv_weight_rows(w.float().cpu().numpy()), b.cpu().numpy(), 25, 31, 320, 18, 3, 1, 1, out_f32=True)
    o = op(nhwc(x))[0]
    assert o.dtype == torch.float32 and tuple(o.shape) == (1, 25, 31, 18)
    assert (o.permute(0, 3, 1, 2) - ref).abs().max().item() < 1e-3 * ref.abs().max().item()


@pytest.mark.parametrize("k,cin", [(1, 128), (2, 128), (4, 256)])
def test_deconv_pixel_shuffle_and_concat(pkg, k, cin):
    """ConvTranspose2d(kernel == stride) + BN bias + ReLU written into a channel slice of a wider tensor
    (the reference's deblocks + concat, src/dsvt-ai-trt.cpp:217-246, 1363)."""
    P = pkg.plugin
    g = torch.Generator(device="cpu").manual_seed(k)
    H, W, cout, total, off = 19, 23, 128, 384, 128
    x = torch.randn(1, cin, H, W, generator=g).half().to(DEV)
    w = (torch.randn(cin, cout, k, k, generator=g) / np.sqrt(cin)).half().to(DEV)
    b = (torch.randn(cout, generator=g) * 0.1).to(DEV)
    ref = torch.relu(F.conv_transpose2d(x.float().cpu(), w.float().cpu(), b.cpu(), stride=k)).to(DEV)      # (CPU reference)
    op = P.add_conv2d_op(P.deconv_weight_rows(w.float().cpu().numpy()), b.cpu().numpy(), H, W, cin, cout, 1, 1, 0,
                         pixel_shuffle=k, relu=True, out_channel_stride=total, out_channel_offset=off)
    buf = torch.full((1, H * k, W * k, total), 7.0, dtype=torch.float16, device=DEV)
    op(nhwc(x), out=[buf])
    torch.cuda.synchronize()
    got = buf[..., off:off + cout].permute(0, 3, 1, 2).float()
    assert (got - ref).abs().max().item() < 2e-3 * ref.abs().max().item()
    assert (buf[..., :off] == 7.0).all() and (buf[..., off + cout:] == 7.0).all()     # other slices untouched


@pytest.mark.parametrize("H,W,cin,cout,k,res", [
    (468, 468, 128, 128, 3, True),      # 885 (8-row) / 705 (10-row) tiles over 256 CUs: every workgroup walks several items
    (234, 234, 64, 320, 3, False),      # three 128-channel chunks per tile, the last one half full
    (117, 117, 256, 256, 1, False),     # 1x1: the resident-weights kernel, one 16-tile column group (two 128-column stages)
    (468, 468, 192, 128, 1, False),     # the first block's shortcut at full size: six k-steps, one stage
    (468, 468, 192, 128, 3, False),     # first BEV conv on the 16-row kernel: six 32-channel phases, 27 slabs
    (468, 468, 64, 384, 3, False),      # head stems on the 16-row kernel: two phases, three chunks, 1350 items
    (250, 200, 128, 128, 3, True),      # 16-row kernel with ragged bottom / right tiles, grid = item count
    (468, 468, 384, 64, 3, False),      # shared head conv: four channel tiles, four steps per weight slab, 27 slabs
    (234, 234, 128, 128, 3, True),      # 8-row x 64-channel tiles on four waves: 480 items, two workgroups per CU
    (117, 117, 256, 256, 3, True),      # ... 240 items, eight phases
    (468, 468, 320, 18, 3, False),      # head outputs: two channel tiles (18 of 32 channels real), 90 steps = 22.5 slabs
])
def test_conv_halo_kernel_full_size(pkg, H, W, cin, cout, k, res):
    """The persistent halo-tile kernel at the BEV sizes (multi-item workgroups, ragged right / bottom tiles)."""
    P = pkg.plugin
    g = torch.Generator(device="cpu").manual_seed(H + cin + cout + k)
    x = torch.randn(1, cin, H, W, generator=g).half().to(DEV)
    w = (torch.randn(cout, cin, k, k, generator=g) / np.sqrt(cin * k * k)).half().to(DEV)
    b = (torch.randn(cout, generator=g) * 0.1).to(DEV)
    r = torch.randn(1, cout, H, W, generator=g).half().to(DEV) if res else None
    ref = _ref(x, w, b, 1, k // 2, r, True)
    op = P.add_conv2d_op(P.conv_weight_rows(w.float().cpu().numpy()), b.cpu().numpy(), H, W, cin, cout, k, 1, k // 2,
                         relu=True, has_residual=res, out_f32=cout % 4 != 0)      # fp16 rows need 8-byte alignment
    args = [nhwc(x)] + ([nhwc(r)] if res else [])
    got = op(*args)[0].permute(0, 3, 1, 2).float()
    torch.cuda.synchronize()
    assert (got - ref).abs().max().item() < 2e-3 * ref.abs().max().item()
    again = op(*args)[0].permute(0, 3, 1, 2).float()
    assert torch.equal(got, again)


@pytest.mark.parametrize("H,W,cin,cout,k,stride,res,relu", [
    (52, 47, 128, 128, 3, 1, True, True),      # basic block conv2 + fp32 identity + ReLU
    (52, 47, 192, 128, 3, 1, False, True),     # first BEV conv
    (52, 47, 128, 256, 3, 2, False, True),     # strided
    (52, 47, 128, 256, 1, 2, False, False),    # 1x1 downsample
    (30, 33, 384, 64, 3, 1, False, True),      # shared head conv (1152 split channels)
    (30, 33, 64, 320, 3, 1, False, True),      # head stems
    (25, 31, 320, 18, 3, 1, False, False),     # head outputs
])
def test_split_precision_conv_is_fp32_grade(pkg, H, W, cin, cout, k, stride, res, relu):
    """fp32 mode of the BEV stage: conv([hi | lo | hi], [w_hi | w_hi | w_lo]) on the fp16 matrix cores (DsvtSplitHalfPlugin +
    plugin.split_weight_rows + DsvtConv2dPlugin with fp32 output) against a float64 convolution of the UNROUNDED fp32 operands
    (convBnLELU / convBn + SUM + ReLU of src/dsvt-ai-trt.cpp:149-246 in fp32): the dropped lo x w_lo terms and the two-step splits
    leave ~2^-21 relative, i.e. fp32 summation-order level."""
    P = pkg.plugin
    g = torch.Generator(device="cpu").manual_seed(H * 1000 + cin + cout)
    x = torch.randn(1, cin, H, W, generator=g) * 3.0
    w = torch.randn(cout, cin, k, k, generator=g) / np.sqrt(cin * k * k)
    b = torch.randn(cout, generator=g) * 0.1
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    r = torch.randn(1, cout, Ho, Wo, generator=g) if res else None
    ref = F.conv2d(x.double(), w.double(), b.double(), stride, pad)
    if res:
        ref = ref + r.double()
    if relu:
        ref = torch.relu(ref)
    rows = P.split_weight_rows(P.conv_weight_rows(w.numpy()), k * k, cin)
    conv = P.add_conv2d_op(rows, b.numpy(), H, W, 3 * cin, cout, k, stride, pad, relu=relu and not res, out_f32=True)
    x32, x3 = P.add_split_half_op(cin)(nhwc(x).to(DEV))
    assert torch.equal(x32, nhwc(x).to(DEV))
    hi, lo = x3[..., :cin].float(), x3[..., cin:2 * cin].float()
    assert torch.equal(x3[..., 2 * cin:], x3[..., :cin]) and (hi + lo - x32).abs().max().item() <= 2.0 ** -21 * x32.abs().max().item()
    y = conv(x3)[0]
    if res:
        y = P.add_split_half_op(cout, relu=relu, has_residual=True)(y, nhwc(r).to(DEV))[0]
    torch.cuda.synchronize()
    got = y.permute(0, 3, 1, 2).double().cpu()
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() < 5e-6 * scale, ((got - ref).abs().max().item(), scale)      # (3456 fp32-accumulated terms at K = 9 x 384)


def test_split_precision_deblock_into_concat(pkg):
    """deconvBnLELU (stride == kernel ConvTranspose, src/dsvt-ai-trt.cpp:217-246) at fp32 grade, written into a channel slice of the fp32
    concat buffer (:1363)"""
    P = pkg.plugin
    g = torch.Generator(device="cpu").manual_seed(11)
    cin, cout, k, H = 128, 128, 2, 26
    x = torch.randn(1, cin, H, H, generator=g)
    w = torch.randn(cin, cout, k, k, generator=g) / np.sqrt(cin)
    b = torch.randn(cout, generator=g) * 0.1
    ref = torch.relu(F.conv_transpose2d(x.double(), w.double(), b.double(), stride=k))
    rows = P.split_weight_rows(P.deconv_weight_rows(w.numpy()), 1, cin)
    op = P.add_conv2d_op(rows, b.numpy(), H, H, 3 * cin, cout, 1, 1, 0, pixel_shuffle=k, relu=True, out_channel_stride=384, out_channel_offset=128,
                         out_f32=True)
    cat = torch.full((1, H * k, H * k, 384), -7.0, dtype=torch.float32, device=DEV)
    op(P.add_split_half_op(cin)(nhwc(x).to(DEV))[1], out=[cat])
    torch.cuda.synchronize()
    got = cat[..., 128:256].permute(0, 3, 1, 2).double().cpu()
    assert (got - ref).abs().max().item() < 5e-6 * ref.abs().max().item()
    assert (cat[..., :128] == -7.0).all() and (cat[..., 256:] == -7.0).all()


@pytest.mark.parametrize("H,W,cin,cout,k,stride,res,shuffle", [
    (468, 468, 128, 128, 3, 1, True, 1),       # 16-row wide kernel, residual
    (234, 234, 128, 128, 3, 1, True, 1),       # a variant chosen by item count: two images change the count
    (117, 117, 256, 256, 3, 1, False, 1),
    (468, 468, 64, 320, 3, 1, False, 1),       # head stems
    (468, 468, 320, 18, 3, 1, False, 1),       # head outputs (fp32, narrow)
    (468, 468, 128, 128, 3, 2, False, 1),      # stride-2 gather kernel (blockIdx.z = image)
    (117, 117, 256, 128, 1, 1, False, 4),      # deblock: pixel shuffle into a channel slice
])
def test_conv_stack_of_images_is_one_launch(pkg, H, W, cin, cout, k, stride, res, shuffle):
    """[B, H, W, C] input: DsvtConv2dPlugin walks the images inside ONE launch (handlesBatch); every image must come out exactly as when
    it is convolved alone."""
    P = pkg.plugin
    g = torch.Generator(device="cpu").manual_seed(H + cin + cout + k + stride)
    B = 3
    x = torch.randn(B, H, W, cin, generator=g).half().to(DEV)
    if shuffle > 1:
        w = torch.randn(cin, cout, shuffle, shuffle, generator=g) / np.sqrt(cin)
        rows = P.deconv_weight_rows(w.numpy())
    else:
        w = torch.randn(cout, cin, k, k, generator=g) / np.sqrt(cin * k * k)
        rows = P.conv_weight_rows(w.numpy())
    b = (torch.randn(cout, generator=g) * 0.1).numpy()
    pad = k // 2
    Ho = ((H + 2 * pad - k) // stride + 1) * shuffle
    kw = dict(relu=True, has_residual=res, out_f32=cout % 4 != 0, pixel_shuffle=shuffle)
    if shuffle > 1:
        kw.update(out_channel_stride=384, out_channel_offset=128)
    r = torch.randn(B, Ho, Ho, cout, generator=g).half().to(DEV) if res else None
    many = P.add_conv2d_op(rows, b, H, W, cin, cout, k, stride, pad, **kw)
    one = P.add_conv2d_op(rows, b, H, W, cin, cout, k, stride, pad, **kw)
    out = many(*([x] + ([r] if res else [])))[0]
    torch.cuda.synchronize()
    assert out.shape[0] == B
    for i in range(B):
        ref = one(*([x[i:i + 1].contiguous()] + ([r[i:i + 1].contiguous()] if res else [])))[0]
        torch.cuda.synchronize()
        sl = slice(128, 256) if shuffle > 1 else slice(None)
        assert torch.equal(out[i][..., sl], ref[0][..., sl]), (i, float((out[i][..., sl].float() - ref[0][..., sl].float()).abs().max()))


@pytest.mark.parametrize("H,W,B", [(468, 468, 1), (61, 45, 3), (7, 5, 2)])
def test_block_diagonal_narrow_conv_on_the_grouped_kernel(pkg, H, W, B):
    """The CenterHead's five output convolutions (64 -> 2 / 1 / 3 / 2 / 10) as ONE 320 -> 18 layer with block-diagonal weights:
    conv3x3_grouped_narrow_kernel (one 64-channel phase per workgroup, weights resident) against PyTorch, and -- bit for bit -- against
    the dense halo kernel, reached by making one structural zero a value that is non-zero in fp32 and zero in fp16."""
    P = pkg.plugin
    g = torch.Generator(device="cpu").manual_seed(H * 7 + W)
    heads = [2, 1, 3, 2, 10]
    cin, cout = 64 * len(heads), sum(heads)
    w = torch.zeros(cout, cin, 3, 3)
    n0 = 0
    for h, n in enumerate(heads):
        w[n0:n0 + n, 64 * h:64 * (h + 1)] = torch.randn(n, 64, 3, 3, generator=g) / 24.0
        n0 += n
    b = torch.randn(cout, generator=g) * 0.1
    x = torch.randn(B, cin, H, W, generator=g).half().to(DEV)
    xin = x.permute(0, 2, 3, 1).contiguous()
    mk = lambda ww: P.add_conv2d_op(P.conv_weight_rows(ww.numpy()), b.numpy(), H, W, cin, cout, 3, 1, 1, out_f32=True)
    got = mk(w)(xin)[0]
    w_dense = w.clone(); w_dense[0, 100, 1, 1] = 1e-30            # channel 0 now "reads" phase 1 as well: dense path, same fp16 weights
    ref_dense = mk(w_dense)(xin)[0]
    torch.cuda.synchronize()
    assert got.dtype == torch.float32 and tuple(got.shape) == (B, H, W, cout)
    assert torch.equal(got, ref_dense), float((got - ref_dense).abs().max())
    ref = _ref(x, w.half().to(DEV), b.to(DEV), 1, 1)
    assert (got.permute(0, 3, 1, 2) - ref).abs().max().item() < 1e-3 * ref.abs().max().item()


def _hi_lo_junk(x):
    """x [B,H,W,C] fp32 (CPU) -> fp16 [B,H,W,3C] = [hi | lo | NaN]: the third plane must never be read (split_input = 1 aliases it to plane 0)"""
    hi = x.half(); lo = (x - hi.float()).half()
    return torch.cat([hi, lo, torch.full_like(hi, float("nan"))], dim=-1).contiguous()


@pytest.mark.parametrize("H,W,cin,cout,k,stride,up,res,relu,B", [
    (52, 47, 192, 128, 1, 1, 1, False, False, 1),     # the 1 x 1 shortcut of the first block: conv1x1_resident_split_kernel<2, 3>
    (468, 468, 128, 128, 1, 1, 1, False, True, 1),    # first deblock at full size: <4, 1>
    (26, 23, 128, 128, 1, 1, 2, False, True, 2),      # second deblock: 2 x 2 pixel shuffle, two images
    (13, 11, 256, 128, 1, 1, 4, False, True, 2),      # third deblock: <4, 2>, sixteen column groups
    (40, 37, 320, 128, 1, 1, 1, False, True, 1),      # a width the resident kernel does not take: the halo kernel
    (468, 468, 128, 128, 3, 1, 1, True, True, 1),     # 16-row x 128-channel items, residual
    (150, 140, 192, 128, 3, 1, 1, False, True, 2),    # six channel groups, two images
    (61, 45, 256, 256, 3, 1, 1, True, True, 1),       # 8-row x 64-channel items (four-step slabs)
    (234, 234, 128, 128, 3, 1, 1, False, True, 1),    # 16-row x 64-channel items
    (150, 140, 384, 64, 3, 1, 1, False, True, 1),     # shared head convolution (64 output channels)
    (150, 140, 64, 320, 3, 1, 1, False, True, 1),     # head stems (two channel groups: the shortest walk)
    (52, 47, 128, 256, 3, 2, 1, False, True, 1),      # strided entry (gather kernel)
])
def test_three_product_layers_over_hi_lo_planes(pkg, H, W, cin, cout, k, stride, up, res, relu, B):
    """The dense stage of the default (three-product) head as pipeline.py wires it since round 5: input [hi | lo | -] with split_input = 1 (the third
    plane's phases alias plane 0; here it holds NaNs), output [hi | lo | -] with split_output = 4 (third plane untouched), residual hi + lo -- against a
    float64 convolution of the unrounded operands (5e-6 of scale: fp32 summation-order level; src/dsvt-ai-trt.cpp:149-246 in fp32).  Covers
    conv_wide_kernel / conv_halo_kernel / conv_f16_kernel with the aliased third plane and conv1x1_resident_split_kernel."""
    P = pkg.plugin
    g = torch.Generator(device="cpu").manual_seed(H * 1000 + cin + cout + k + up)
    x = torch.randn(B, H, W, cin, generator=g) * 3.0
    x = torch.relu(x) if (H + cin) % 2 else x
    if up > 1:
        w = torch.randn(cin, cout, up, up, generator=g) / np.sqrt(cin)
        rows = P.deconv_weight_rows(w.numpy())
        ref = F.conv_transpose2d(x.permute(0, 3, 1, 2).double(), w.double(), None, stride=up)
    else:
        w = torch.randn(cout, cin, k, k, generator=g) / np.sqrt(cin * k * k)
        rows = P.conv_weight_rows(w.numpy())
        ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), None, stride, k // 2)
    b = torch.randn(cout, generator=g) * 0.1
    ref = ref + b.double()[None, :, None, None]
    Ho, Wo = ref.shape[2], ref.shape[3]
    r = torch.randn(B, Ho, Wo, cout, generator=g) if res else None
    if res:
        r3 = _hi_lo_junk(r)
        ref = ref + (r3[..., :cout].double() + r3[..., cout:2 * cout].double()).permute(0, 3, 1, 2)
    if relu:
        ref = torch.relu(ref)
    plane, off = (384, 128) if up > 1 else (cout, 0)
    op = P.add_conv2d_op(P.split_weight_rows(rows, k * k, cin), b.numpy(), H, W, 3 * cin, cout, k, stride, k // 2, pixel_shuffle=up, relu=relu, has_residual=res,
                         split_residual=1 if res else 0, split_input=1, split_output=4, out_channel_stride=3 * plane, out_channel_offset=off)
    out = torch.full((B, Ho, Wo, 3 * plane), 7.0, dtype=torch.float16, device=DEV)
    args = [_hi_lo_junk(x).to(DEV)] + ([r3.to(DEV)] if res else [])
    op(*args, out=[out])
    torch.cuda.synchronize()
    o = out.cpu()
    got = (o[..., off:off + cout].double() + o[..., plane + off:plane + off + cout].double()).permute(0, 3, 1, 2)
    assert not torch.isnan(got).any()
    scale = ref.abs().max().item()
    err = (got - ref).abs().max().item() / scale
    print(f"three-product {H}x{W} {cin}->{cout} k{k} s{stride} up{up}: {err:.2e} of scale")
    assert err < 5e-6, err
    untouched = torch.ones(3 * plane, dtype=torch.bool)
    untouched[off:off + cout] = False; untouched[plane + off:plane + off + cout] = False
    assert (o[..., untouched] == 7.0).all()                       # the third plane (and the neighbours of a concat slice) are left alone
    again = torch.full_like(out, 7.0)
    op(*args, out=[again]); torch.cuda.synchronize()
    assert torch.equal(again.view(torch.int16), out.view(torch.int16))


def test_shared_head_conv_on_24_row_items_equals_its_16_row_form(pkg):
    """The 64-output-channel three-product layer (shared 384 -> 64 head convolution) picks its item height by the round count: four 468 x 468 images take
    24-row items (three rows per wave: conv_wide_kernel<4, 8, 36, 4, 2, 3>), one image 16-row ones.  Both walk the same K order, so the stack's images
    must be the BITS of the single-image launches; and image 0 is checked against a float64 convolution on a crop."""
    P = pkg.plugin
    g = torch.Generator(device="cpu").manual_seed(77)
    H, cin, cout, B = 468, 384, 64, 4
    w = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(cin * 9)
    b = torch.randn(cout, generator=g) * 0.1
    x = torch.relu(torch.randn(B, H, H, cin, generator=g))
    x3 = _hi_lo_junk(x).to(DEV)
    kw = dict(relu=True, split_input=1, split_output=4, out_channel_stride=3 * cout)
    rows = P.split_weight_rows(P.conv_weight_rows(w.numpy()), 9, cin)
    many = P.add_conv2d_op(rows, b.numpy(), H, H, 3 * cin, cout, 3, 1, 1, **kw)
    one = P.add_conv2d_op(rows, b.numpy(), H, H, 3 * cin, cout, 3, 1, 1, **kw)
    out = many(x3)[0]
    torch.cuda.synchronize()
    for i in range(B):
        ref = one(x3[i:i + 1].contiguous())[0]
        torch.cuda.synchronize()
        assert torch.equal(out[i][..., :2 * cout].view(torch.int16), ref[0][..., :2 * cout].view(torch.int16)), i
    # a 40 x 468 band of image 0 (rows 200 .. 239: item boundaries of both heights inside) against float64
    y0, y1 = 200, 240
    crop = x[0:1, y0 - 1:y1 + 1].permute(0, 3, 1, 2).double()
    ref = torch.relu(F.conv2d(crop, w.double(), b.double(), 1, (0, 1)))[0].permute(1, 2, 0)
    got = (out[0, y0:y1, :, :cout].double() + out[0, y0:y1, :, cout:2 * cout].double()).cpu()
    assert (got - ref).abs().max().item() < 5e-6 * ref.abs().max().item()


@pytest.mark.parametrize("H,W,cin,cout,res,B,split_out", [
    (468, 468, 128, 128, False, 1, 4),     # the BEV ResNet layer (14 of the frame's launches): 450 items on 256 CUs, the last round partial
    (468, 468, 128, 128, True, 2, 4),      # ... with a residual, two images: items walk image after image
    (468, 468, 128, 128, True, 4, 4),      # ... four images: 1800 items would be 7 rounds + 8; the partial last tile row (four rows) is a second launch of 8-row x 64-channel items
    (468, 468, 192, 128, False, 1, 1),     # the first block's entry: 18 phases; [hi | lo | hi] output
    (150, 140, 128, 256, True, 4, 4),      # two channel chunks per tile, image edges inside the last tile column / row
    (117, 117, 256, 256, False, 4, 4),     # the third stage at four frames: exactly 256 items
    (200, 190, 64, 320, False, 2, 4),      # the head stems' shape, small: six phases, five 64-channel chunks on 24-row items (conv_rows_kernel<4, 3>; waves 0-3 request two weight rows, 4-7 one)
    (468, 468, 64, 320, False, 1, 4),      # ... at full size
    (468, 468, 384, 64, False, 3, 4),      # the shared head convolution, three images: 36 phases, ONE 64-channel chunk, 24-row items (two images would take 16-row ones)
    (468, 468, 384, 64, False, 1, 4),      # ... one image: 450 16-row items (conv_rows_kernel<4, 2>; round 5: conv_wide_kernel<4, 8, 40, 4, 2, 2>)
    (234, 234, 128, 128, True, 1, 4),      # the second stage at one frame: 240 items of 16 rows x 64 channels (<4, 2>), fewer items than workgroups
    (117, 117, 256, 256, True, 2, 4),      # the third stage at two frames: 256 items of 16 rows x 64 channels (<4, 2>)
    (117, 117, 256, 256, False, 1, 4),     # ... at one frame: 240 items of 8 rows x 64 channels (conv_rows_kernel<4, 1>; round 5: conv_wide_kernel<4, 8, 36, 4, 2, 1>)
    (117, 117, 256, 256, True, 3, 4),      # ... at three frames: 720 8-row items, three rounds
    (40, 50, 64, 64, False, 1, 1),         # a small image: 10 items of 8 rows, a grid of 10 workgroups, [hi | lo | hi] output
])
def test_rows_kernel_equals_wide_kernel(pkg, H, W, cin, cout, res, B, split_out):
    """conv_rows_kernel<8, 2> / <4, 3> / <4, 2> / <4, 1> (round 6, csrc/conv_rows.hip: ky-row slabs, requests through buffer descriptors, 34-pixel halo rows) against round 5's
    conv_wide_kernel<8, 8, 36, 4, 2, 2, SPL> / <4, 8, 36, 4, 2, 3, SPL> / <4, 8, 40, 4, 2, 2, SPL> / <4, 8, 36, 4, 2, 1, SPL> on the same three-product layer: both walk the (phase, tap) steps in the same order into the same accumulators,
    so every output bit must agree (kernel_variant = 1 keeps a layer on the round-5 kernel); and against a float64 convolution (5e-6 of scale)."""
    P = pkg.plugin
    g = torch.Generator(device="cpu").manual_seed(H * 7 + cin + cout + B)
    x = torch.relu(torch.randn(B, H, W, cin, generator=g) * 3.0)
    w = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(cin * 9)
    b = torch.randn(cout, generator=g) * 0.1
    rows = P.split_weight_rows(P.conv_weight_rows(w.numpy()), 9, cin)
    r3 = _hi_lo_junk(torch.randn(B, H, W, cout, generator=g)) if res else None
    outs = []
    for variant in (0, 1):
        op = P.add_conv2d_op(rows, b.numpy(), H, W, 3 * cin, cout, 3, 1, 1, relu=True, has_residual=res, split_residual=1 if res else 0, split_input=1,
                             split_output=split_out, out_channel_stride=3 * cout, kernel_variant=variant)
        out = torch.full((B, H, W, 3 * cout), 7.0, dtype=torch.float16, device=DEV)
        args = [_hi_lo_junk(x).to(DEV)] + ([r3.to(DEV)] if res else [])
        op(*args, out=[out])
        op(*args, out=[out])                                          # (a second launch into the same buffers: nothing carried over)
        torch.cuda.synchronize()
        outs.append(out.cpu())
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16)), int((outs[0].view(torch.int16) != outs[1].view(torch.int16)).sum())
    ref = F.conv2d(x[:1].permute(0, 3, 1, 2).double(), w.double(), b.double(), 1, 1)
    if res:
        ref = ref + (r3[:1, ..., :cout].double() + r3[:1, ..., cout:2 * cout].double()).permute(0, 3, 1, 2)
    ref = torch.relu(ref)
    o = outs[0][:1]
    got = (o[..., :cout].double() + o[..., cout:2 * cout].double()).permute(0, 3, 1, 2)
    assert (got - ref).abs().max().item() < 5e-6 * ref.abs().max().item()
