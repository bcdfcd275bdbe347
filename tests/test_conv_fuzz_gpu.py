"""Random small shapes through the convolution kernels that keep their weights resident in LDS (round 2): the 1 x 1 streaming GEMM
(stride 1 / 2, pixel shuffle), the 64-input-channel 3 x 3 kernel and the block-diagonal narrow 3 x 3 kernel -- images smaller than a
tile, ragged right / bottom tiles, stacks of images -- against PyTorch's fp32 CPU convolution on the same fp16 operands (the reference is computed
on the host: until round 4 it ran on the GPU, i.e. MIOpen checked these kernels)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def _cases(seed, n):
    rng = np.random.default_rng(seed)
    return [tuple(int(v) for v in (rng.integers(1, 70), rng.integers(1, 90), rng.integers(1, 4), rng.integers(0, 1 << 30))) for _ in range(n)]


@pytest.mark.parametrize("H,W,B,seed", _cases(11, 24))
def test_fuzz_conv1x1_resident(pkg, H, W, B, seed):
    P = pkg.plugin
    rng = np.random.default_rng(seed)
    cin = int(rng.choice([128, 192, 256])); stride = int(rng.choice([1, 1, 2])); up = 1 if stride == 2 else int(rng.choice([1, 2, 4]))
    cout = 128 if up > 1 else int(rng.choice([128, 256]))
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(B, cin, H, W, generator=g).half().to(DEV)
    b = (torch.randn(cout, generator=g) * 0.1)
    if up > 1:
        w = torch.randn(cin, cout, up, up, generator=g) / np.sqrt(cin)
        ref = torch.relu(F.conv_transpose2d(x.float().cpu(), w.half().float(), b, stride=up)).to(DEV)
        op = P.add_conv2d_op(P.deconv_weight_rows(w.numpy()), b.numpy(), H, W, cin, cout, 1, 1, 0, pixel_shuffle=up, relu=True)
    else:
        w = torch.randn(cout, cin, 1, 1, generator=g) / np.sqrt(cin)
        ref = torch.relu(F.conv2d(x.float().cpu(), w.half().float(), b, stride)).to(DEV)
        op = P.add_conv2d_op(P.conv_weight_rows(w.numpy()), b.numpy(), H, W, cin, cout, 1, stride, 0, relu=True)
    got = op(_nhwc(x))[0]
    torch.cuda.synchronize()
    assert tuple(got.shape) == (B, ref.shape[2], ref.shape[3], cout)
    assert (got.permute(0, 3, 1, 2).float() - ref).abs().max().item() < 2e-3 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("H,W,B,seed", _cases(12, 20))
def test_fuzz_conv3x3_c64_resident(pkg, H, W, B, seed):
    P = pkg.plugin
    rng = np.random.default_rng(seed)
    cout = int(rng.choice([64, 128, 320]))
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(B, 64, H, W, generator=g).half().to(DEV)
    w = torch.randn(cout, 64, 3, 3, generator=g) / 24.0
    b = torch.randn(cout, generator=g) * 0.1
    ref = torch.relu(F.conv2d(x.float().cpu(), w.half().float(), b, 1, 1)).to(DEV)
    got = P.add_conv2d_op(P.conv_weight_rows(w.numpy()), b.numpy(), H, W, 64, cout, 3, 1, 1, relu=True)(_nhwc(x))[0]
    torch.cuda.synchronize()
    assert (got.permute(0, 3, 1, 2).float() - ref).abs().max().item() < 2e-3 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("H,W,B,seed", _cases(13, 20))
def test_fuzz_block_diagonal_narrow_conv(pkg, H, W, B, seed):
    P = pkg.plugin
    rng = np.random.default_rng(seed)
    nph = int(rng.integers(2, 6))
    heads = [int(rng.integers(1, 13)) for _ in range(nph)]
    order = rng.permutation(nph)                               # the phases need not come in channel order
    cin, cout = 64 * nph, sum(heads)
    g = torch.Generator(device="cpu").manual_seed(seed)
    w = torch.zeros(cout, cin, 3, 3)
    n0 = 0
    for h, n in enumerate(heads):
        ph = int(order[h])
        w[n0:n0 + n, 64 * ph:64 * (ph + 1)] = torch.randn(n, 64, 3, 3, generator=g) / 24.0
        n0 += n
    b = torch.randn(cout, generator=g) * 0.1
    x = torch.randn(B, cin, H, W, generator=g).half().to(DEV)
    ref = F.conv2d(x.float().cpu(), w.half().float(), b, 1, 1).to(DEV)
    got = P.add_conv2d_op(P.conv_weight_rows(w.numpy()), b.numpy(), H, W, cin, cout, 3, 1, 1, out_f32=True)(_nhwc(x))[0]
    torch.cuda.synchronize()
    assert got.dtype == torch.float32
    assert (got.permute(0, 3, 1, 2) - ref).abs().max().item() < 1e-3 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("H,W,B,seed", _cases(14, 12))
def test_fuzz_conv3x3_wide_kernels_with_residual(pkg, H, W, B, seed):
    """the 3 x 3 "phase" kernels (whichever variant the item count picks) with a residual input: the epilogue loads the residual rows of
    eight blocks before their stores"""
    P = pkg.plugin
    rng = np.random.default_rng(seed)
    H, W = 2 * H, 2 * W                                        # up to 138 x 178: from one tile to several rounds of items
    cin = int(rng.choice([128, 192, 256])); cout = int(rng.choice([128, 256]))
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(B, cin, H, W, generator=g).half().to(DEV)
    w = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(9 * cin)
    b = torch.randn(cout, generator=g) * 0.1
    r = torch.randn(B, cout, H, W, generator=g).half().to(DEV)
    ref = torch.relu(F.conv2d(x.float().cpu(), w.half().float(), b, 1, 1) + r.float().cpu()).to(DEV)
    got = P.add_conv2d_op(P.conv_weight_rows(w.numpy()), b.numpy(), H, W, cin, cout, 3, 1, 1, relu=True, has_residual=True)(_nhwc(x), _nhwc(r))[0]
    torch.cuda.synchronize()
    assert (got.permute(0, 3, 1, 2).float() - ref).abs().max().item() < 2e-3 * max(1.0, ref.abs().max().item())


def test_rows_kernel_fuzz_against_the_round5_kernel(pkg):
    """conv_rows_kernel<8, 2> / <4, 3> on random three-product layers that reach them (whole-chip grids: enough images), against round 5's kernels of the same layers
    (kernel_variant = 1) bit for bit: odd heights / widths (partial last tile row and column, images that end inside a tile), every phase count from 6 to 24, channel
    counts with a half-empty last chunk, residuals, one to several channel chunks, item counts that are and are not a multiple of the grid (the start skew)."""
    import torch
    P = pkg.plugin
    rng = np.random.default_rng(2024)
    done = {"rows": 0, "rows64": 0}
    for trial in range(10):
        cin = int(rng.choice([64, 128, 192, 256]))
        cout = int(rng.choice([64, 128, 192, 256, 320]))
        H, W = int(rng.integers(97, 230)), int(rng.integers(97, 230))
        res = bool(rng.integers(0, 2))
        ct4 = cout <= 64 or cout % 128 == 64
        tiles = -(-H // (24 if ct4 else 16)) * -(-W // 32) * -(-cout // (64 if ct4 else 128))
        B = max(1, -(-256 // tiles)) + int(rng.integers(0, 2))
        g = torch.Generator(device="cpu").manual_seed(1000 + trial)
        x = torch.relu(torch.randn(B, H, W, cin, generator=g) * 2.0)
        hi = x.half(); x3 = torch.cat([hi, (x - hi.float()).half(), torch.full_like(hi, float("nan"))], -1).to(DEV)      # (the third plane aliases plane 0: never read)
        w = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(cin * 9)
        b = torch.randn(cout, generator=g) * 0.1
        rows = P.split_weight_rows(P.conv_weight_rows(w.numpy()), 9, cin)
        r3 = None
        if res:
            r = torch.randn(B, H, W, cout, generator=g); rh = r.half()
            r3 = torch.cat([rh, (r - rh.float()).half(), torch.full_like(rh, float("nan"))], -1).to(DEV)
        outs = []
        for variant in (0, 1):
            op = P.add_conv2d_op(rows, b.numpy(), H, W, 3 * cin, cout, 3, 1, 1, relu=bool(trial % 3), has_residual=res, split_residual=1 if res else 0, split_input=1,
                                 split_output=4, out_channel_stride=3 * cout, kernel_variant=variant)
            out = torch.full((B, H, W, 3 * cout), 7.0, dtype=torch.float16, device=DEV)
            op(*([x3] + ([r3] if res else [])), out=[out])
            torch.cuda.synchronize()
            outs.append(out.cpu())
        a, c = outs[0].view(torch.int16), outs[1].view(torch.int16)
        assert torch.equal(a, c), (trial, H, W, cin, cout, B, res, int((a != c).sum()))
        assert not torch.isnan(outs[0][..., :2 * cout].float()).any() and (outs[0][..., 2 * cout:] == 7.0).all()
        done["rows64" if ct4 else "rows"] += 1
    assert done["rows"] >= 2 and done["rows64"] >= 2, done


def test_small_rows_kernels_fuzz_against_the_round5_kernels(pkg):
    """conv_rows_kernel<4, 2> / <4, 1> (the launches that do not fill the chip: few images, small maps) and the two-launch form of a 128-channel layer whose partial last
    tile row saves a round of the chip (csrc/conv_rows.hip launchRows128), against round 5's kernels of the same layers (kernel_variant = 1) bit for bit: one to three images,
    maps from a single tile up, odd sizes, 64 .. 384 output channels, residuals."""
    import torch
    P = pkg.plugin
    rng = np.random.default_rng(77)
    shapes = [(20, 1024, 128, 256, 4, True),       # 1 full tile row + 4 rows, 512 items: 256 on <8, 2> + the four rows on <4, 1> (two launches)
              (36, 500, 64, 128, 8, False),        # 2 full tile rows + 4 rows, 384 items: 256 + 128 small ones
              (9, 30, 64, 64, 1, False)]           # one item
    for trial in range(9):
        cin = int(rng.choice([64, 128, 256])); cout = int(rng.choice([64, 128, 192, 256, 384]))
        shapes.append((int(rng.integers(17, 120)), int(rng.integers(17, 120)), cin, cout, int(rng.integers(1, 4)), bool(rng.integers(0, 2))))
    for trial, (H, W, cin, cout, B, res) in enumerate(shapes):
        g = torch.Generator(device="cpu").manual_seed(5000 + trial)
        x = torch.relu(torch.randn(B, H, W, cin, generator=g) * 2.0)
        hi = x.half(); x3 = torch.cat([hi, (x - hi.float()).half(), torch.full_like(hi, float("nan"))], -1).to(DEV)
        w = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(cin * 9)
        b = torch.randn(cout, generator=g) * 0.1
        rows = P.split_weight_rows(P.conv_weight_rows(w.numpy()), 9, cin)
        r3 = None
        if res:
            r = torch.randn(B, H, W, cout, generator=g); rh = r.half()
            r3 = torch.cat([rh, (r - rh.float()).half(), torch.full_like(rh, float("nan"))], -1).to(DEV)
        outs = []
        for variant in (0, 1):
            op = P.add_conv2d_op(rows, b.numpy(), H, W, 3 * cin, cout, 3, 1, 1, relu=bool(trial % 2), has_residual=res, split_residual=1 if res else 0, split_input=1,
                                 split_output=4, out_channel_stride=3 * cout, kernel_variant=variant)
            out = torch.full((B, H, W, 3 * cout), 7.0, dtype=torch.float16, device=DEV)
            op(*([x3] + ([r3] if res else [])), out=[out])
            torch.cuda.synchronize()
            outs.append(out.cpu())
        a, c = outs[0].view(torch.int16), outs[1].view(torch.int16)
        assert torch.equal(a, c), (trial, H, W, cin, cout, B, res, int((a != c).sum()))
        assert not torch.isnan(outs[0][..., :2 * cout].float()).any() and (outs[0][..., 2 * cout:] == 7.0).all()
        # and against float64 on image 0
        ref = F.conv2d(x[:1].permute(0, 3, 1, 2).double(), w.double(), b.double(), 1, 1)
        if res:
            ref = ref + (r3[:1, ..., :cout].cpu().double() + r3[:1, ..., cout:2 * cout].cpu().double()).permute(0, 3, 1, 2)
        if trial % 2:
            ref = torch.relu(ref)
        o = outs[0][:1]
        got = (o[..., :cout].double() + o[..., cout:2 * cout].double()).permute(0, 3, 1, 2)
        assert (got - ref).abs().max().item() < 5e-6 * max(1.0, ref.abs().max().item()), (trial, H, W, cin, cout)
