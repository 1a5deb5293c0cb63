"""Operator-level parity: every C entry point against the CPU oracle's op (stock torch CPU)
on the same seeded inputs.  fp32 kernels must agree to fp32 rounding; bf16 kernels are
compared with the oracle evaluated on the bf16-rounded operands, so the only difference
left is accumulation order and the final bf16 rounding of stored activations."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import unet_oracle as O

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]
# relL2 bars: fp32 = accumulation-order noise; bf16 = one rounding of the stored result
TOL = {torch.float32: 2e-6, torch.bfloat16: 4e-3}
TOL_W = {torch.float32: 5e-6, torch.bfloat16: 4e-3}   # fp32-accumulated, fp32-stored weight gradients


@pytest.fixture(scope="module")
def ua():
    import unet_amd
    unet_amd.lib()
    return unet_amd


def rel(a, b):
    return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()


def rnd(dtype, t):
    """Round to the storage dtype and back (what the kernels see)."""
    return t.to(dtype).float()


def gen(*shape, seed=0, scale=1.0, shift=0.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) - shift) * scale


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(3, 20, 24, 9, 7, 0), (2, 12, 32, 16, 16, 0), (2, 40, 8, 5, 6, 24),
                                   (5, 64, 64, 2, 2, 0), (1, 3, 70, 11, 13, 0),
                                   # tiny images: position-major kernels (batch tails, non-square, concat, both K splits)
                                   (70, 128, 64, 4, 4, 0), (3, 128, 128, 2, 4, 64), (600, 64, 128, 4, 4, 0),
                                   (130, 64, 64, 1, 3, 0),
                                   # 4x4 images, 16 per halo tile (the wide-halo variant of the tile kernel)
                                   (32, 64, 96, 4, 4, 0), (48, 32, 16, 4, 4, 0), (16, 128, 64, 4, 4, 64),
                                   # wide images: two 128-pixel rows per halo tile
                                   (2, 32, 64, 4, 128, 0), (1, 64, 32, 6, 128, 32)])
def test_conv3x3_fwd_dgrad_wgrad(ua, dtype, shape):
    _check_conv3x3(ua, dtype, shape)


@pytest.mark.parametrize("shape", [(70, 256, 128, 4, 4, 128), (600, 128, 256, 2, 2, 0), (9, 128, 128, 1, 3, 0),
                                   (33, 128, 256, 4, 2, 0), (512, 256, 256, 4, 4, 0), (1, 128, 128, 4, 4, 0),
                                   # the 5-level network's bottleneck (models/unet2d.py:37-42 with features[-1]=512:
                                   # 512 -> 1024 and 1024 -> 1024, K = 9216) and its first decoder conv (concat 512+512)
                                   (8, 512, 1024, 4, 4, 0), (8, 1024, 1024, 4, 4, 0), (6, 1024, 512, 8, 8, 512),
                                   (40, 1024, 1024, 2, 2, 0)])
def test_conv3x3_tiny_images_wide_channels(ua, shape):
    """bf16, images of at most 16 pixels with the channel counts of the deepest levels: the per-tap
    weight-gradient reduction (ragged pixel ranges, one and many splits, concat input, 1-row images)."""
    _check_conv3x3(ua, torch.bfloat16, shape)


@pytest.mark.parametrize("shape", [(4, 512, 1024, 4, 4, 0), (4, 1024, 1024, 4, 4, 0), (2, 1024, 512, 8, 8, 512)])
def test_conv3x3_1024_channels_fp32(ua, shape):
    """The same 1024-channel layers through the exact-fp32 kernels."""
    _check_conv3x3(ua, torch.float32, shape)


@pytest.mark.parametrize("shape", [(1, 128, 256, 8, 8, 0), (1, 256, 256, 8, 8, 0), (1, 512, 256, 8, 8, 256),
                                   (1, 256, 512, 4, 4, 0), (1, 64, 64, 4, 4, 0), (8, 128, 256, 8, 8, 0),
                                   (3, 64, 48, 8, 8, 0), (1, 32, 32, 5, 7, 0), (2, 96, 32, 16, 16, 0),
                                   (1, 64, 32, 32, 32, 32), (1, 512, 512, 4, 4, 0)])
@pytest.mark.parametrize("dtype", DTYPES)
def test_conv3x3_few_pixels(ua, dtype, shape):
    """Layers of one to a few images at the deep levels (the shapes of a batch-1 forward of the
    64x64 sweep), bf16 and fp32: the K-split kernel for layers with few pixels, one and two pixel
    fragments per workgroup, four and eight K waves, ragged pixel counts, padded channels, concat
    input, a chunk count that is not a power of two."""
    _check_conv3x3(ua, dtype, shape)


def _geometry_sweep():
    """Every power-of-two image size the tile kernels can see, batch chosen for at least two 256-pixel tiles."""
    shapes = []
    for H in (2, 4, 8, 16, 32, 64):
        for W in (4, 8, 16, 32, 64, 128):
            if H * W > 4096:
                continue
            B = max(2, 512 // (H * W))
            shapes.append((B, 32, 64, H, W, 0))
    return shapes


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", _geometry_sweep())
def test_conv3x3_image_geometries(ua, dtype, shape):
    _check_conv3x3(ua, dtype, shape)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("hw", [(32, 32), (8, 8), (4, 4), (2, 2)])
def test_conv3x3_dgrad_relu_mask_edge_values(ua, dtype, hw):
    """The input gradient through a ReLU (models/unet2d.py:30, :40) survives exactly where the stored activation is
    > 0: +0, -0 and negative values block it, the smallest subnormal and +inf pass it (the bf16 kernels decide on the
    activation's bit pattern, two values per operation).  Compared with the unmasked gradient value by value."""
    ops = ua.ops
    H, W = hw
    B, Cin, Cout = 8, 64, 32
    dev = "cuda"
    w = rnd(dtype, gen(Cout, Cin, 3, 3, seed=2, shift=0.5, scale=0.2))
    gy = rnd(dtype, gen(B, Cout, H, W, seed=4, shift=0.5))
    _, wb = ops.pack_conv3x3(w.to(dev), dtype)
    dyn = ops.to_nhwc(gy.to(dev), dtype)
    plain = ops.conv3x3_dgrad(dyn, wb, Cin, Cout)
    tiny = torch.finfo(dtype).smallest_normal * (2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -23)   # smallest subnormal
    values = torch.tensor([0.0, -0.0, tiny, -tiny, 1.5, -1.5, float("inf"), -float("inf"), 3e-39, 65504.0], dtype=torch.float32)
    idx = torch.randint(0, len(values), (B, H, W, plain.shape[3]), generator=torch.Generator().manual_seed(7))
    mask = values[idx].to(dtype).to(dev)
    assert (mask.float() == values[idx].to(dtype).float().to(dev)).all()
    masked = ops.conv3x3_dgrad(dyn, wb, Cin, Cout, mask=mask)
    keep = mask.float() > 0
    assert keep.any() and (~keep).any()
    assert torch.equal(masked[keep], plain[keep])
    assert torch.count_nonzero(masked[~keep]).item() == 0


def _check_conv3x3(ua, dtype, shape):
    B, Cin, Cout, H, W, split = shape
    ops = ua.ops
    x = rnd(dtype, gen(B, Cin, H, W, seed=1, shift=0.3))
    w = rnd(dtype, gen(Cout, Cin, 3, 3, seed=2, shift=0.5, scale=0.2))
    b = gen(Cout, seed=3, shift=0.5)
    xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    pre = F.conv2d(xr, wr, br, padding=1)
    y_ref = torch.relu(pre)
    dev = "cuda"
    # concat layout: channels [0,split) then padding, then the rest
    if split:
        xa = ops.to_nhwc(x[:, :split].to(dev), dtype)
        xb = ops.to_nhwc(x[:, split:].to(dev), dtype)
        xn = torch.cat([xa, xb], dim=3).contiguous()
    else:
        xn = ops.to_nhwc(x.to(dev), dtype)
    wf, wb = ops.pack_conv3x3(w.to(dev), dtype, split=split)
    yn = ops.conv3x3_relu_fwd(xn, wf, b.to(dev), Cin, Cout, split=split)
    y = ops.from_nhwc(yn, Cout).cpu()
    assert rel(y, y_ref.detach()) < TOL[dtype]
    assert torch.count_nonzero(yn[..., Cout:]).item() == 0          # padding channels stay zero

    # backward: dy is the gradient w.r.t. the pre-activation (mask already applied)
    gy = rnd(dtype, gen(B, Cout, H, W, seed=4, shift=0.5) * (y_ref.detach() > 0))
    pre.backward(gy)
    dyn = ops.to_nhwc(gy.to(dev), dtype)
    dxn = ops.conv3x3_dgrad(dyn, wb, Cin, Cout, split=split)
    if split:
        sp = ops.pad_ch(split)
        dx = torch.cat([ops.from_nhwc(dxn, split), ops.from_nhwc(dxn, Cin - split, coff=sp)], dim=1).cpu()
    else:
        dx = ops.from_nhwc(dxn, Cin).cpu()
    assert rel(dx, xr.grad) < TOL[dtype]
    # fused ReLU mask of the preceding layer
    if not split:
        mask_src = rnd(dtype, gen(B, Cin, H, W, seed=5, shift=0.5))
        dxm = ops.from_nhwc(ops.conv3x3_dgrad(dyn, wb, Cin, Cout, mask=ops.to_nhwc(mask_src.to(dev), dtype)), Cin).cpu()
        assert rel(dxm, xr.grad * (mask_src > 0)) < TOL[dtype]
    dW, db = ops.conv3x3_wgrad(xn, dyn, Cin, Cout, split=split)
    assert rel(dW.cpu(), wr.grad) < TOL_W[dtype]
    assert rel(db.cpu(), br.grad) < TOL_W[dtype]
    # alpha/beta accumulate form
    dW2, db2 = dW.clone(), db.clone()
    ops.conv3x3_wgrad(xn, dyn, Cin, Cout, split=split, dW=dW2, db=db2, alpha=0.5, beta=1.0)
    assert rel(dW2.cpu(), 1.5 * wr.grad) < TOL_W[dtype] and rel(db2.cpu(), 1.5 * br.grad) < TOL_W[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv3x3_large_pixel_count(ua, dtype):
    """Enough pixels to take the 256-pixel tile path and many weight-gradient splits."""
    ops = ua.ops
    B, Cin, Cout, H, W = 96, 32, 32, 32, 32
    x = rnd(dtype, gen(B, Cin, H, W, seed=11, shift=0.4))
    w = rnd(dtype, gen(Cout, Cin, 3, 3, seed=12, shift=0.5, scale=0.1))
    b = gen(Cout, seed=13, shift=0.5)
    xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    pre = F.conv2d(xr, wr, br, padding=1)
    y_ref = torch.relu(pre).detach()
    gy = rnd(dtype, gen(B, Cout, H, W, seed=14, shift=0.5) * (y_ref > 0))
    pre.backward(gy)
    xn = ops.to_nhwc(x.cuda(), dtype)
    wf, wb = ops.pack_conv3x3(w.cuda(), dtype)
    y = ops.from_nhwc(ops.conv3x3_relu_fwd(xn, wf, b.cuda(), Cin, Cout), Cout).cpu()
    assert rel(y, y_ref) < TOL[dtype]
    dyn = ops.to_nhwc(gy.cuda(), dtype)
    assert rel(ops.from_nhwc(ops.conv3x3_dgrad(dyn, wb, Cin, Cout), Cin).cpu(), xr.grad) < TOL[dtype]
    dW, db = ops.conv3x3_wgrad(xn, dyn, Cin, Cout)
    assert rel(dW.cpu(), wr.grad) < TOL_W[dtype] * 4 and rel(db.cpu(), br.grad) < TOL_W[dtype] * 4
    dW_b, _ = ops.conv3x3_wgrad(xn, dyn, Cin, Cout)
    assert torch.equal(dW, dW_b)                                   # fixed-order reduction: bitwise repeatable


@pytest.mark.parametrize("dtype", DTYPES)
def test_maxpool_known_answers(ua, dtype, golden):
    ops = ua.ops
    g = golden("pool_ties")
    for name in ("even", "odd"):
        x, y, gy, gx = (torch.from_numpy(g[f"{name}.{k}"]) for k in ("x", "y", "gy", "gx"))
        if dtype == torch.bfloat16:   # keep the tie structure: values are exactly representable or re-derived
            x = rnd(dtype, x)
            xr = x.clone().requires_grad_(True)
            y = O.maxpool2(xr); gy = rnd(dtype, gy); y.backward(gy); gx = xr.grad; y = y.detach()
        C = x.shape[1]
        xn = ops.to_nhwc(x.cuda(), dtype)
        yn = ops.maxpool2_fwd(xn, C)
        assert torch.equal(ops.from_nhwc(yn, C).cpu(), y)           # a max is exact
        gxn = ops.maxpool2_bwd(xn, ops.to_nhwc(gy.cuda(), dtype), C)
        assert torch.equal(ops.from_nhwc(gxn, C).cpu(), gx)         # routing to the FIRST maximum, exact
        # fused form: + skip gradient, * relu mask
        sk = rnd(dtype, gen(*x.shape, seed=9, shift=0.5))
        fused = ops.from_nhwc(ops.maxpool2_bwd(xn, ops.to_nhwc(gy.cuda(), dtype), C,
                                               dskip=ops.to_nhwc(sk.cuda(), dtype), relu_mask=True), C).cpu()
        assert rel(fused, rnd(dtype, (gx + sk) * (x > 0))) < 1e-7


@pytest.mark.parametrize("shape", [(3, 64, 8, 12), (2, 256, 4, 4), (2, 32, 32, 32), (5, 128, 2, 6), (1, 40, 6, 4), (2, 512, 2, 2)])
def test_maxpool_bwd_bf16_even_images(ua, shape):
    """bf16, even H and W, channel counts from one 16-byte group per pixel up to 64 of them: values quantised so
    that most windows hold ties, exact routing to the first maximum, with and without the fused skip gradient."""
    ops = ua.ops
    B, C, H, W = shape
    dtype = torch.bfloat16
    x = torch.round(gen(B, C, H, W, seed=11, shift=0.5) * 6) / 4          # few distinct values: many ties
    xr = x.clone().requires_grad_(True)
    y = O.maxpool2(xr)
    gy = rnd(dtype, gen(*y.shape, seed=12, shift=0.5))
    y.backward(gy)
    xn = ops.to_nhwc(x.cuda(), dtype)
    gxn = ops.maxpool2_bwd(xn, ops.to_nhwc(gy.cuda(), dtype), C)
    assert torch.equal(ops.from_nhwc(gxn, C).cpu(), xr.grad)
    sk = rnd(dtype, gen(*x.shape, seed=13, shift=0.5))
    fused = ops.from_nhwc(ops.maxpool2_bwd(xn, ops.to_nhwc(gy.cuda(), dtype), C, dskip=ops.to_nhwc(sk.cuda(), dtype),
                                           relu_mask=True), C).cpu()
    assert torch.equal(fused, rnd(dtype, (xr.grad + sk) * (x > 0)))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(3, 32, 16, 5, 7), (2, 64, 32, 8, 8), (2, 8, 4, 3, 3), (4, 512, 256, 2, 2),
                                   (3, 128, 64, 7, 5), (2, 64, 24, 6, 6),
                                   # the 5-level network's first transposed conv (models/unet2d.py:46-48, 1024 -> 512)
                                   (2, 1024, 512, 4, 4), (64, 1024, 512, 4, 4),
                                   # >= 65536 pixels: the streaming kernel (weights in LDS), with a ragged last block
                                   (67, 64, 32, 32, 31), (70, 128, 64, 31, 31), (66, 64, 24, 32, 32),
                                   # fewer pixel blocks than CUs / a weight matrix beyond 64 KB: the same kernel cut
                                   # along the output channels (2 and 8 slices)
                                   (33, 128, 64, 32, 31), (40, 256, 128, 16, 15)])
def test_convt2x2(ua, dtype, shape):
    B, Cin, Cout, H, W = shape
    ops = ua.ops
    x = rnd(dtype, gen(B, Cin, H, W, seed=1, shift=0.3))
    w = rnd(dtype, gen(Cin, Cout, 2, 2, seed=2, shift=0.5, scale=0.2))
    b = gen(Cout, seed=3, shift=0.5)
    xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    y_ref = O.convt2x2(xr, wr, br)
    gy = rnd(dtype, gen(B, Cout, 2 * H, 2 * W, seed=4, shift=0.5))
    y_ref.backward(gy)
    xn = ops.to_nhwc(x.cuda(), dtype)
    wf, wb = ops.pack_convt2x2(w.cuda(), dtype)
    # write into the second half of a concat buffer, as the decoder does
    cp = ops.pad_ch(Cout)
    cat = torch.zeros((B, 2 * H, 2 * W, 2 * cp), device="cuda", dtype=dtype)
    ops.convt2x2_fwd(xn, wf, b.cuda(), Cin, Cout, out=cat, out_coff=cp)
    assert rel(ops.from_nhwc(cat, Cout, coff=cp).cpu(), y_ref.detach()) < TOL[dtype]
    assert torch.count_nonzero(cat[..., :cp]).item() == 0
    dcat = torch.zeros_like(cat)
    ops.to_nhwc(gy.cuda(), dtype, out=dcat, coff=cp)
    mask_src = rnd(dtype, gen(B, Cin, H, W, seed=5, shift=0.5))
    dx = ops.from_nhwc(ops.convt2x2_dgrad(dcat, wb, Cin, Cout, dy_coff=cp), Cin).cpu()
    assert rel(dx, xr.grad) < TOL[dtype]
    dxm = ops.from_nhwc(ops.convt2x2_dgrad(dcat, wb, Cin, Cout, mask=ops.to_nhwc(mask_src.cuda(), dtype), dy_coff=cp), Cin).cpu()
    assert rel(dxm, xr.grad * (mask_src > 0)) < TOL[dtype]
    dW, db = ops.convt2x2_wgrad(xn, dcat, Cin, Cout, dy_coff=cp)
    # the bias gradient sums B*2H*2W signed values: at 265 K terms the fp32 reference sum itself is only good
    # to ~1e-5, so it is held to the gradient bar (5e-5) rather than to the operator tolerance
    assert rel(dW.cpu(), wr.grad) < TOL_W[dtype] and rel(db.cpu(), br.grad) < max(TOL_W[dtype], 5e-5)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cout", [1, 12])
def test_head(ua, dtype, cout):
    ops = ua.ops
    B, Cin, H, W = 5, 32, 13, 9
    x = rnd(dtype, gen(B, Cin, H, W, seed=1, shift=0.4))
    w = gen(cout, Cin, 1, 1, seed=2, shift=0.5)
    b = gen(cout, seed=3, shift=0.5)
    xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    y_ref = O.conv1x1(xr, wr, br)
    gy = gen(B, cout, H, W, seed=4, shift=0.5)
    y_ref.backward(gy)
    xn = ops.to_nhwc(x.cuda(), dtype)
    pred = ops.head_fwd(xn, w.cuda(), b.cuda(), Cin, cout)
    assert rel(pred.cpu(), y_ref.detach()) < 2e-6                   # fp32 accumulate and output in both modes
    dx, dW, db = ops.head_bwd(xn, w.cuda(), gy.cuda(), Cin, cout, relu_mask=True)
    assert rel(ops.from_nhwc(dx, Cin).cpu(), xr.grad * (x > 0)) < TOL[dtype]
    assert rel(dW.cpu(), wr.grad) < 5e-6 and rel(db.cpu(), br.grad) < 5e-6


def test_mse(ua):
    ops = ua.ops
    for n_shape in [(7, 1, 5, 3), (64, 1, 32, 32), (3, 12, 17, 9)]:
        p = gen(*n_shape, seed=1); y = gen(*n_shape, seed=2)
        pr = p.clone().requires_grad_(True)
        loss = O.mse_loss(pr, y); loss.backward()
        out3, dp = ops.mse_fwd_bwd(p.cuda(), y.cuda())
        out3 = out3.cpu()
        assert abs(out3[0].item() - loss.item()) <= 2e-7 * abs(loss.item()) + 1e-12
        d = (p - y).double()
        assert abs(out3[1].item() - d.abs().sum().item()) <= 1e-6 * d.abs().sum().item()
        assert abs(out3[2].item() - (d ** 2).sum().item()) <= 1e-6 * (d ** 2).sum().item()
        assert rel(dp.cpu(), pr.grad) < 2e-7
        gs = torch.tensor([0.25], device="cuda")
        _, dp2 = ops.mse_fwd_bwd(p.cuda(), y.cuda(), gscale_dev=gs)
        assert rel(dp2.cpu(), 0.25 * pr.grad) < 2e-7
        none3, dp3 = ops.mse_fwd_bwd(p.cuda(), y.cuda(), gscale_dev=gs, want_sums=False)   # gradient only (the loss's backward)
        assert none3 is None and torch.equal(dp3, dp2)
    with pytest.raises(ValueError):
        ops.mse_fwd_bwd(torch.zeros(2, 3, device="cuda"), torch.zeros(3, 2, device="cuda"))


@pytest.mark.parametrize("sizes", [((4, 4), (5, 5)), ((26, 24), (27, 24)), ((27, 24), (15, 24)), ((2, 2), (3, 3)),
                                   ((12, 12), (13, 12)), ((10, 10), (11, 11)), ((8, 8), (8, 8)), ((7, 9), (3, 4))])
def test_bilinear(ua, sizes):
    ops = ua.ops
    (hi, wi), (ho, wo) = sizes
    x = gen(2, 5, hi, wi, seed=1)
    xr = x.clone().requires_grad_(True)
    y_ref = O.bilinear(xr, (ho, wo))
    gy = gen(2, 5, ho, wo, seed=2, shift=0.5)
    y_ref.backward(gy)
    y = ops.bilinear_fwd_nchw(x.cuda(), (ho, wo)).cpu()
    assert rel(y, y_ref.detach()) < 2e-7
    gx = ops.bilinear_bwd_nchw(gy.cuda(), (hi, wi)).cpu()
    assert rel(gx, xr.grad) < 5e-7


def test_adam_matches_torch(ua):
    ops = ua.ops
    n = 1000 * 7 + 3
    p0 = gen(n, seed=1, shift=0.5)
    p_ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p_ref], lr=1e-4)                       # train.py:497
    p = p0.cuda().clone(); m = torch.zeros_like(p); v = torch.zeros_like(p)
    for step in range(1, 6):
        g = gen(n, seed=10 + step, shift=0.5, scale=0.01)
        p_ref.grad = g.clone()
        opt.step()
        ops.adam_step(p, g.cuda(), m, v, 1e-4, 0.9, 0.999, 1e-8, step)
    assert rel(p.cpu() - p0, p_ref.detach() - p0) < 1e-5
    assert rel(p.cpu(), p_ref.detach()) < 1e-7


def test_layout_roundtrip_and_errors(ua):
    ops = ua.ops
    x = gen(3, 12, 5, 7, seed=1)
    for dtype in DTYPES:
        t = ops.to_nhwc(x.cuda(), dtype)
        assert t.shape == (3, 5, 7, ops.pad_ch(12)) and torch.count_nonzero(t[..., 12:]).item() == 0
        assert torch.equal(ops.from_nhwc(t, 12).cpu(), rnd(dtype, x))
        x2 = gen(2, 40, 4, 6, seed=2)         # 48 stored channels: a partial second block of 32
        t2 = ops.to_nhwc(x2.cuda(), dtype)
        assert t2.shape == (2, 4, 6, 48) and torch.count_nonzero(t2[..., 40:]).item() == 0
        assert torch.equal(ops.from_nhwc(t2, 40).cpu(), rnd(dtype, x2))
    with pytest.raises(RuntimeError):
        ops.to_nhwc(x, torch.float32)          # CPU tensor: no fallback
    # argument errors surface as RuntimeError with the library's message
    bad = torch.zeros((1, 4, 4, 24), device="cuda")  # channel stride not a multiple of 16
    wf, _ = ops.pack_conv3x3(torch.zeros(16, 24, 3, 3, device="cuda"), torch.float32)
    with pytest.raises(RuntimeError, match="multiples of 16"):
        ops.conv3x3_relu_fwd(bad, wf, torch.zeros(16, device="cuda"), 24, 16)
