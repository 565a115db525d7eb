"""Per-kernel parity (-m gpu): every C-ABI entry point against the torch-CPU op the reference calls
(the oracle's arithmetic), on seeded inputs, plus the committed known-answer fixtures (g4_ops.npz).

Tolerances: fp32 accumulation-order noise only -- rel-L2 <= 2e-6 for GEMM-like ops, max-abs scaled
<= 1e-5; integer outputs (pool indices) bit-exact.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def nhwc(t):      # NCHW cpu -> NHWC gpu
    return t.permute(0, 2, 3, 1).contiguous().to(dev())


def nchw(t):      # NHWC gpu -> NCHW cpu
    return t.permute(0, 3, 1, 2).contiguous().cpu()


def rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def close(a, b, tol=2e-6, name=""):
    r = rel_l2(a, b)
    assert r <= tol, f"{name}: rel-L2 {r:.3e} > {tol}"
    scale = float(b.abs().max()) + 1e-30
    m = float((a.double() - b.double()).abs().max()) / scale
    assert m <= 50 * tol, f"{name}: max-abs/scale {m:.3e}"


CONV_SHAPES = [  # n, h, w, cin, cout
    (2, 8, 16, 8, 24),        # ragged channels, 64x64 tile path, edge predication
    (1, 4, 4, 4, 4),          # minimum
    (8, 64, 64, 32, 128),     # 128x128 tiles
    (8, 64, 64, 64, 64),      # 128x64 tiles
    (2, 16, 16, 128, 256),    # multi K-chunk per tap, 64x64 path
    (3, 32, 32, 20, 36),      # non-power-of-two channels
    (16, 64, 64, 20, 36),     # halo-reuse kernel <64>: ragged output channels, 16-channel chunk tail
    (16, 64, 64, 24, 132),    # halo-reuse kernel <128>: ragged N tile (132 of 256 columns), chunk tail
    (2, 32, 32, 32, 132),     # strip weight gradient with a masked output-channel block (132 = 128 + 4)
    (2, 16, 32, 160, 32),     # strip weight gradient, swapped roles (Cout < 128 <= Cin), non-square image
    (1, 8, 16, 64, 128),      # smallest image the strip kernel takes (one 16-pixel strip, 8 rows)
    (4, 16, 16, 128, 128),    # 16x16 level: several strips per block
    (1, 256, 16, 32, 128),    # one 16-pixel strip, 8 row chunks per image (strip kernel split along y)
    (1, 64, 64, 64, 256),     # 4 strips x 2 row chunks, two output-channel tiles
    (2, 128, 128, 32, 128),   # halo kernel on a small batch (N = 128: one column tile), strip kernel with chunks
    # tiles that are not powers of two (fast-division pixel indexing; lib/UNet.py is fully convolutional)
    (2, 24, 40, 32, 64),      # generic kernels only (W % 16 != 0)
    (3, 12, 20, 8, 24),       # odd quotients everywhere
    (2, 48, 80, 64, 128),     # W % 16 == 0, H % 8 == 0: halo kernel + strip weight gradient on a 48 x 80 image
    (1, 6, 48, 32, 128),      # strip kernel with 6 rows (even, not a multiple of 4: one chunk); H % 8 != 0: no halo kernel
    (16, 40, 96, 24, 132),    # halo kernel <128> with 5 x 6 patches per image, ragged N tile
    (4, 32, 32, 160, 32),     # strip weight gradient <*,4,1> with swapped operand roles (Cout < 128 <= Cin, Cin % 64 != 0)
    (4, 32, 32, 96, 160),     # ... un-swapped with a ragged second row tile (Cout = 128 + 32)
    (2, 16, 32, 192, 320),    # square-tile strip kernel <*,2,2>: 5 x 3 tiles, two strips per image row
    (32, 8, 8, 512, 512),     # the cfg-S bottleneck (8 x 8 images: two-image patches; weight gradient on image-PAIR strips, r04)
    (33, 8, 8, 64, 256),      # ... with an odd image count (the last pair lacks its second image) and Cin != Cout
    (1, 8, 8, 64, 64),        # a single 8 x 8 image: one half-empty pair
    (5, 8, 8, 128, 192),      # three pairs, 3 x 2 tiles
]


def _random_conv_shapes():
    import random
    rnd = random.Random(20260928)
    out = []
    for _ in range(14):
        h = rnd.choice([8, 16, 32, 64]); w = rnd.choice([8, 16, 32, 64, 128])
        n = rnd.choice([1, 2, 3, 5, 8])
        cin = 4 * rnd.randint(1, 48); cout = 4 * rnd.randint(1, 48)
        if rnd.random() < 0.4:
            cin = rnd.choice([32, 64, 96, 128, 160])
        if rnd.random() < 0.4:
            cout = rnd.choice([128, 132, 192, 256])
        if n * h * w * (cin + cout) > 6e6:      # keep the torch-CPU reference quick
            n = 1
        out.append((n, h, w, cin, cout))
    return out


@pytest.mark.parametrize("n,h,w,cin,cout", _random_conv_shapes())
def test_conv3x3_random_shapes(n, h, w, cin, cout):
    """Seeded random shapes across the tile / kernel selection rules (generic split kernel, halo kernel, strip and TN
    weight-gradient kernels, swapped roles, ragged channel counts)."""
    test_conv3x3_fwd_dgrad_wgrad(n, h, w, cin, cout)



@pytest.mark.parametrize("n,h,w,cin,cout", CONV_SHAPES)
def test_conv3x3_fwd_dgrad_wgrad(n, h, w, cin, cout):
    from resdepth_amd import ops
    g = torch.Generator().manual_seed(n * 1000 + cin)
    x = torch.randn(n, cin, h, w, generator=g, requires_grad=True)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)).requires_grad_(True)
    gy = torch.randn(n, cout, h, w, generator=g)
    y = F.conv2d(x, wt, None, 1, 1)
    y.backward(gy)
    wf, wd = ops.pack_conv3x3_weight(wt.detach().to(dev()))
    z = ops.conv3x3_fwd(nhwc(x.detach()), wf)
    close(nchw(z), y.detach(), name="fwd")
    dx = ops.conv3x3_bwd_data(nhwc(gy), wd)
    close(nchw(dx), x.grad, name="dgrad")
    dw = ops.conv3x3_bwd_weight(nhwc(x.detach()), nhwc(gy))
    close(dw.cpu(), wt.grad, name="wgrad")


FULL_LAYERS = [  # cfg-S layers at the BASELINE batch (N=32): name, h, cin, cout -- no CPU reference at this size
    ("enc1", 128, 64, 128), ("enc3", 32, 256, 512), ("dec2", 64, 256, 128), ("dec3", 128, 128, 64), ("enc4", 16, 512, 512)]


@pytest.mark.parametrize("name,h,cin,cout", FULL_LAYERS)
def test_conv3x3_adjoint_identities_and_linearity_at_full_size(name, h, cin, cout):
    """Size-independent properties at BASELINE.json's full layer sizes (halo / strip kernels, several strips per block):
    <conv(x), g> = <x, dgrad(g)> = <w, wgrad(x, g)>  (forward, data gradient and weight gradient are mutually adjoint)
    and conv(a*x1 + b*x2) = a*conv(x1) + b*conv(x2).  Dot products in fp64 on the device."""
    from resdepth_amd import ops
    n = 32
    gen = torch.Generator(device=dev()).manual_seed(h * 7 + cin)
    x = torch.randn(n, h, h, cin, device=dev(), generator=gen)
    x2 = torch.randn(n, h, h, cin, device=dev(), generator=gen)
    g = torch.randn(n, h, h, cout, device=dev(), generator=gen)
    w = torch.randn(cout, cin, 3, 3, device=dev(), generator=gen) / (3 * cin ** 0.5)
    wf, wd = ops.pack_conv3x3_weight(w)
    z = ops.conv3x3_fwd(x, wf)
    dx = ops.conv3x3_bwd_data(g, wd)
    dw = ops.conv3x3_bwd_weight(x, g)
    dot = lambda a, b: float((a.double() * b.double()).sum())
    ref = dot(z, g)
    scale = float(z.double().norm() * g.double().norm())
    assert abs(dot(x, dx) - ref) <= 1e-5 * scale, (name, ref, dot(x, dx))
    assert abs(dot(w, dw) - ref) <= 1e-5 * scale, (name, ref, dot(w, dw))
    z12 = ops.conv3x3_fwd(0.75 * x - 1.5 * x2, wf)
    z2 = ops.conv3x3_fwd(x2, wf)
    lin = 0.75 * z - 1.5 * z2
    assert rel_l2(z12, lin) <= 2e-6, (name, rel_l2(z12, lin))


@pytest.mark.parametrize("h,c", [(128, 64), (64, 128), (32, 256), (16, 512)])
def test_convt2x2_adjoint_identities_at_full_size(h, c):
    """<convT(x) , g> = <x, dgrad(g)> = <w, wgrad(x, g)> at the cfg-S decoder sizes, batch 32 (bias and skip off)."""
    from resdepth_amd import ops
    n = 32
    gen = torch.Generator(device=dev()).manual_seed(h + c)
    x = torch.randn(n, h, h, c, device=dev(), generator=gen)
    g = torch.randn(n, 2 * h, 2 * h, c, device=dev(), generator=gen)
    w = torch.randn(c, c, 2, 2, device=dev(), generator=gen) / (2 * c ** 0.5)
    wtf, wtd = ops.pack_convt2x2_weight(w)
    out = ops.convt2x2_fwd(x, wtf, None, None)
    dx = ops.convt2x2_bwd_data(g, wtd)
    dw = ops.convt2x2_bwd_weight(x, g)
    dot = lambda a, b: float((a.double() * b.double()).sum())
    ref = dot(out, g)
    scale = float(out.double().norm() * g.double().norm())
    assert abs(dot(x, dx) - ref) <= 1e-5 * scale, (ref, dot(x, dx))
    assert abs(dot(w, dw) - ref) <= 1e-5 * scale, (ref, dot(w, dw))


CONVT_SHAPES = [(2, 4, 8, 8, 12), (1, 2, 2, 4, 4), (4, 16, 16, 128, 128), (8, 32, 32, 64, 64), (2, 8, 8, 256, 256),
                (2, 6, 10, 8, 12), (3, 12, 48, 64, 64), (2, 24, 40, 128, 128), (1, 5, 8, 32, 64),     # not powers of two
                # r03 kernels of rd_convt.hip: convt_dgrad<2,1,4> (Cin >= 128) / <2,2,2> (Cin = 64), several column tiles, a
                # ragged last pixel tile, Cin != Cout; convt_wgrad<4|2|1> with an ODD number of K-steps per split (the extra
                # all-zero step), several row tiles, a ragged last split
                (4, 32, 32, 128, 128), (2, 48, 48, 256, 128), (1, 80, 80, 64, 192), (3, 36, 48, 128, 64), (5, 30, 32, 192, 64),
                (1, 65, 70, 64, 128), (1, 67, 68, 128, 64)]      # ragged last pixel tile of convt_dgrad (M = 4550 / 4556)


@pytest.mark.parametrize("n,h,w,cin,cout", CONVT_SHAPES)
def test_convt2x2_fwd_dgrad_wgrad(n, h, w, cin, cout):
    from resdepth_amd import ops
    g = torch.Generator().manual_seed(n * 77 + cin)
    x = torch.randn(n, cin, h, w, generator=g, requires_grad=True)
    wt = (torch.randn(cin, cout, 2, 2, generator=g) / cin ** 0.5).requires_grad_(True)
    b = torch.randn(cout, generator=g, requires_grad=True)
    skip = torch.randn(n, cout, 2 * h, 2 * w, generator=g)
    gy = torch.randn(n, cout, 2 * h, 2 * w, generator=g)
    y = skip + F.conv_transpose2d(x, wt, b, stride=2)
    y.backward(gy)
    wtf, wtd = ops.pack_convt2x2_weight(wt.detach().to(dev()))
    out = ops.convt2x2_fwd(nhwc(x.detach()), wtf, b.detach().to(dev()), nhwc(skip))
    close(nchw(out), y.detach(), name="fwd")
    out2 = ops.convt2x2_fwd(nhwc(x.detach()), wtf, None, None)
    close(nchw(out2), F.conv_transpose2d(x.detach(), wt.detach(), None, stride=2), name="fwd-nobias")
    dx = ops.convt2x2_bwd_data(nhwc(gy), wtd)
    close(nchw(dx), x.grad, name="dgrad")
    dw = ops.convt2x2_bwd_weight(nhwc(x.detach()), nhwc(gy))
    close(dw.cpu(), wt.grad, name="wgrad")
    db = ops.channel_sum(nhwc(gy))
    close(db.cpu(), b.grad, tol=2e-5, name="dbias")   # torch-CPU fp32 sum is the noisy side (ours is fp64)


def test_convt_known_answer(g4):
    from resdepth_amd import ops
    t = lambda k: torch.from_numpy(g4["convt/" + k].copy())
    wtf, wtd = ops.pack_convt2x2_weight(t("w").to(dev()))
    # fixture has cin=4, cout=6 (cout not a multiple of 4 -> only the forward and wgrad-free parts apply)
    out = ops.convt2x2_fwd(nhwc(t("x")), wtf, t("b").to(dev()), None)
    close(nchw(out), t("y"), name="convt KA fwd")


def test_conv_known_answer(g4):
    from resdepth_amd import ops
    t = lambda k: torch.from_numpy(g4["conv/" + k].copy())
    wf, wd = ops.pack_conv3x3_weight(t("w").to(dev()))
    close(nchw(ops.conv3x3_fwd(nhwc(t("x")), wf)), t("y"), name="conv KA fwd")
    close(nchw(ops.conv3x3_bwd_data(nhwc(t("gy")), wd)), t("gx"), name="conv KA dgrad")
    close(ops.conv3x3_bwd_weight(nhwc(t("x")), nhwc(t("gy"))).cpu(), t("gw"), name="conv KA wgrad")


@pytest.mark.parametrize("n,h,w,cin,cout", [(2, 32, 32, 3, 8), (1, 16, 16, 1, 4), (3, 8, 64, 2, 64), (2, 64, 64, 3, 64),
                                            (1, 16, 16, 6, 12),
                                            # segment kernels (Cin <= 4, Cout = 32 / 64 / 128): ragged tiles, every Cin
                                            (1, 40, 72, 4, 32), (2, 16, 32, 1, 128), (1, 256, 256, 3, 64), (3, 24, 8, 2, 64),
                                            (2, 32, 32, 5, 64)])
def test_first_conv(n, h, w, cin, cout):
    from resdepth_amd import ops
    g = torch.Generator().manual_seed(cin * 10 + cout)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / 3).requires_grad_(True)
    gy = torch.randn(n, cout, h, w, generator=g)
    y = F.conv2d(x, wt, None, 1, 1)
    y.backward(gy)
    z = ops.conv3x3_first_fwd(x.to(dev()), wt.detach().to(dev()))
    close(nchw(z), y.detach(), name="first fwd")
    dw = ops.conv3x3_first_bwd_weight(x.to(dev()), nhwc(gy))
    close(dw.cpu(), wt.grad, tol=5e-6, name="first wgrad")


@pytest.mark.parametrize("n,h,w,c,xc,bias", [(2, 32, 32, 8, 3, True), (1, 16, 16, 4, 1, False), (2, 64, 64, 64, 3, True),
                                             (3, 8, 8, 20, 2, True),
                                             # tile kernels (C = 16 / 32 / 64): ragged tiles in both directions, one-tile images
                                             (3, 40, 72, 64, 3, True), (1, 8, 8, 64, 1, False), (2, 24, 100, 32, 2, True),
                                             (5, 16, 32, 16, 1, True), (1, 256, 256, 64, 3, True)])
def test_last_conv(n, h, w, c, xc, bias):
    from resdepth_amd import ops
    g = torch.Generator().manual_seed(c)
    s = torch.randn(n, c, h, w, generator=g, requires_grad=True)
    wt = (torch.randn(1, c, 3, 3, generator=g) / (3 * c ** 0.5)).requires_grad_(True)
    b = torch.randn(1, generator=g, requires_grad=True) if bias else None
    x = torch.randn(n, xc, h, w, generator=g)
    gy = torch.randn(n, 1, h, w, generator=g)
    y = x[:, 0:1] + F.conv2d(s, wt, b, 1, 1)
    y.backward(gy)
    out = ops.conv3x3_last_fwd(nhwc(s.detach()), wt.detach().to(dev()), b.detach().to(dev()) if bias else None,
                               x.to(dev()))
    close(out.cpu(), y.detach(), name="last fwd")
    ds = ops.conv3x3_last_bwd_data(gy.to(dev()), wt.detach().to(dev()), c)
    close(nchw(ds), s.grad, name="last dgrad")
    dw, db = ops.conv3x3_last_bwd_weight(nhwc(s.detach()), gy.to(dev()), want_bias=bias)
    close(dw.cpu(), wt.grad, tol=5e-6, name="last wgrad")
    if bias:
        close(db.cpu(), b.grad, tol=5e-6, name="last dbias")


@pytest.mark.parametrize("n,h,w,c,slope,pool", [(3, 8, 8, 8, 0.0, True), (2, 16, 32, 64, 0.0, True),
                                                (2, 16, 16, 12, 0.01, True), (4, 8, 8, 512, 0.0, False),
                                                (2, 4, 4, 32, 0.01, False),
                                                (2, 12, 20, 8, 0.0, True), (3, 6, 10, 12, 0.01, True), (1, 24, 40, 64, 0.0, False)])
def test_bn_act_pool_forward_backward(n, h, w, c, slope, pool):
    from resdepth_amd import ops
    g = torch.Generator().manual_seed(c + n)
    x = (torch.randn(n, c, h, w, generator=g) * 1.7 + 0.6).requires_grad_(True)
    gam = (torch.rand(c, generator=g) + 0.5).requires_grad_(True)
    bet = (torch.randn(c, generator=g) * 0.3).requires_grad_(True)
    rm, rv = torch.randn(c, generator=g), torch.rand(c, generator=g) + 0.5
    rm0, rv0 = rm.clone(), rv.clone()
    y = F.batch_norm(x, rm, rv, gam, bet, True, 0.1, 1e-5)
    a = F.leaky_relu(y, slope) if slope else F.relu(y)
    if pool:
        p, idx = F.max_pool2d(a, 2, 2, return_indices=True)
        g_pool = torch.randn(p.shape, generator=g)
        g_full = torch.randn(a.shape, generator=g)
        (p * g_pool).sum().add((a * g_full).sum()).backward()
    else:
        g_full = torch.randn(a.shape, generator=g)
        (a * g_full).sum().backward()
    z = nhwc(x.detach())
    drm, drv = rm0.to(dev()), rv0.to(dev())
    nbt = torch.zeros((), dtype=torch.long, device=dev())
    sums = ops.bn_stats_partial(z)
    mean, invstd = ops.bn_stats_finalize(sums, n * h * w, drm, drv, nbt)
    assert int(nbt) == 1
    np.testing.assert_allclose(drm.cpu().numpy(), rm.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(drv.cpu().numpy(), rv.numpy(), rtol=1e-5, atol=1e-6)
    ga, be = gam.detach().to(dev()), bet.detach().to(dev())
    da, dp, didx = ops.bn_act_pool_fwd(z, mean, invstd, ga, be, slope, pool)
    close(nchw(da), a.detach(), tol=2e-6, name="bn act")
    if pool:
        # bit-exact pooling given identical inputs: pool OUR activation with torch and compare
        p2, idx2 = F.max_pool2d(nchw(da), 2, 2, return_indices=True)
        assert torch.equal(nchw(dp), p2)
        iy, ix = idx2 // w, idx2 % w
        pos = (iy % 2) * 2 + (ix % 2)
        assert torch.equal(nchw(didx).long(), pos)
    gf = nhwc(g_full)
    gp = nhwc(g_pool) if pool else None
    bs = ops.bn_act_bwd_reduce(z, mean, invstd, ga, be, slope, gf, gp, didx)
    dgam = torch.empty(c, device=dev())
    dbet = torch.empty(c, device=dev())
    dz = ops.bn_act_bwd_apply(z, mean, invstd, ga, be, slope, gf, gp, didx, bs, n * h * w, True, dgam, dbet)
    close(nchw(dz), x.grad, tol=2e-5, name="bn dz")
    close(dgam.cpu(), gam.grad, tol=1e-5, name="dgamma")
    close(dbet.cpu(), bet.grad, tol=1e-5, name="dbeta")
    close(bs[2 * c:3 * c].float().cpu(), g_full.sum((0, 2, 3)), tol=1e-5, name="sum g_full")
    # eval mode: running statistics, constants in the backward
    emean, einv = ops.bn_eval_stats(drm, drv)
    ea, _, _ = ops.bn_act_pool_fwd(z, emean, einv, ga, be, slope, False)
    ye = F.batch_norm(x.detach(), rm, rv, gam.detach(), bet.detach(), False, 0.1, 1e-5)
    close(nchw(ea), F.leaky_relu(ye, slope) if slope else F.relu(ye), tol=2e-6, name="bn eval")


def test_pool_ties_and_nan_known_answer(g4):
    """first maximum in row-major window order; NaN wins (golden from torch's MaxPool2d)."""
    from resdepth_amd import ops
    x = torch.from_numpy(g4["pool/x"].copy())              # [1,1,4,8]
    xx = x.repeat(1, 4, 1, 1)                              # 4 channels (kernels want C % 4 == 0)
    z = nhwc(xx)
    one, zero = torch.ones(4, device=dev()), torch.zeros(4, device=dev())
    a, p, idx = ops.bn_act_pool_fwd(z, zero, one, one, zero, 1.0, True)   # slope 1 => identity activation
    ref_p = torch.from_numpy(g4["pool/y"])
    ref_i = torch.from_numpy(g4["pool/idx"]).long()
    pos = ((ref_i // 8) % 2) * 2 + (ref_i % 8) % 2
    got_p = nchw(p)[:, 0:1]
    assert torch.equal(torch.isnan(got_p), torch.isnan(ref_p))
    assert torch.equal(torch.nan_to_num(got_p, nan=7.0), torch.nan_to_num(ref_p, nan=7.0))
    assert torch.equal(nchw(idx)[:, 0:1].long(), pos)
    # backward routing on the random relu case
    x2 = torch.from_numpy(g4["pool2/x"].copy())            # [2,5,8,8]
    x2 = torch.cat([x2, x2[:, :3]], 1)                     # 8 channels
    gy = torch.from_numpy(g4["pool2/gy"].copy())
    gy = torch.cat([gy, gy[:, :3]], 1)
    gx = torch.from_numpy(g4["pool2/gx"].copy())
    gx = torch.cat([gx, gx[:, :3]], 1)
    one, zero = torch.ones(8, device=dev()), torch.zeros(8, device=dev())
    z2 = nhwc(x2)
    a, p, idx = ops.bn_act_pool_fwd(z2, zero, one, one, zero, 1.0, True)
    sums = ops.bn_act_bwd_reduce(z2, zero, one, one, zero, 1.0, None, nhwc(gy), idx)
    dz = ops.bn_act_bwd_apply(z2, zero, one, one, zero, 1.0, None, nhwc(gy), idx, sums, 1.0, False)
    assert torch.equal(nchw(dz), gx)


@pytest.mark.parametrize("pool", [False, True])
def test_prelu_learnable_slope_on_device(pool):
    """nn.PReLU() (lib/UNet.py:29): the slope is read from device memory and its gradient is the 4th reduced quantity."""
    from resdepth_amd import ops
    g = torch.Generator().manual_seed(77)
    n, h, w, c = 2, 16, 16, 8
    x = torch.randn(n, c, h, w, generator=g, requires_grad=True)
    gam = (torch.rand(c, generator=g) + 0.5).requires_grad_(True)
    bet = (torch.randn(c, generator=g) * 0.3).requires_grad_(True)
    sl = torch.tensor([0.25], requires_grad=True)
    y = F.batch_norm(x, None, None, gam, bet, True, 0.1, 1e-5)
    a = F.prelu(y, sl)
    g_full = torch.randn(a.shape, generator=g)
    obj = (a * g_full).sum()
    g_pool = None
    if pool:
        p = F.max_pool2d(a, 2, 2)
        g_pool = torch.randn(p.shape, generator=g)
        obj = obj + (p * g_pool).sum()
    obj.backward()
    z = nhwc(x.detach())
    mean, invstd = ops.bn_stats_finalize(ops.bn_stats_partial(z), n * h * w, torch.zeros(c, device=dev()),
                                         torch.ones(c, device=dev()), torch.zeros((), dtype=torch.long, device=dev()))
    ga, be, dsl = gam.detach().to(dev()), bet.detach().to(dev()), sl.detach().to(dev())
    da, dp, didx = ops.bn_act_pool_fwd(z, mean, invstd, ga, be, 0.0, pool, slope_dev=dsl)
    close(nchw(da), a.detach(), tol=2e-6, name="prelu act")
    gf, gp = nhwc(g_full), (nhwc(g_pool) if pool else None)
    bs = ops.bn_act_bwd_reduce(z, mean, invstd, ga, be, 0.0, gf, gp, didx, slope_dev=dsl)
    dgam, dbet = torch.empty(c, device=dev()), torch.empty(c, device=dev())
    dz = ops.bn_act_bwd_apply(z, mean, invstd, ga, be, 0.0, gf, gp, didx, bs, n * h * w, True, dgam, dbet, slope_dev=dsl)
    close(nchw(dz), x.grad, tol=2e-5, name="prelu dz")
    close(dgam.cpu(), gam.grad, tol=1e-5, name="prelu dgamma")
    close(bs[3 * c:].sum().float().cpu().reshape(1), sl.grad, tol=1e-5, name="d slope")


@pytest.mark.parametrize("shape", [(2, 8, 8, 8), (1, 1, 1, 4), (3, 4, 16, 12), (1, 2, 1, 8)])
def test_bilinear_upsample_matches_torch_and_adjoint(shape):
    """nn.Upsample(scale_factor=2, mode='bilinear') (lib/UNet.py:20) + bias + skip, and its adjoint."""
    from resdepth_amd import ops
    n, h, w, c = shape
    g = torch.Generator().manual_seed(5)
    t = torch.randn(n, c, h, w, generator=g, requires_grad=True)
    bias = torch.randn(c, generator=g)
    skip = torch.randn(n, c, 2 * h, 2 * w, generator=g)
    ref = F.interpolate(t, scale_factor=2, mode="bilinear") + bias.view(1, -1, 1, 1) + skip
    gy = torch.randn(ref.shape, generator=g)
    (ref * gy).sum().backward()
    out = ops.upsample2x_add_fwd(nhwc(t.detach()), bias.to(dev()), nhwc(skip))
    close(nchw(out), ref.detach(), tol=1e-6, name="upsample fwd")
    dt = ops.upsample2x_bwd(nhwc(gy))
    close(nchw(dt), t.grad, tol=1e-6, name="upsample adjoint")


def test_conv1x1_coarse_grid_equals_reference_order():
    """conv1x1(upsample(x)) (reference order) == upsample(conv1x1(x)) (engine order) to fp32 rounding."""
    from resdepth_amd import ops
    g = torch.Generator().manual_seed(6)
    n, c, h, w = 2, 24, 16, 16
    x = torch.randn(n, c, h, w, generator=g, requires_grad=True)
    conv = torch.nn.Conv2d(c, c, 1)
    skip = torch.randn(n, c, 2 * h, 2 * w, generator=g)
    ref = conv(F.interpolate(x, scale_factor=2, mode="bilinear")) + skip
    gy = torch.randn(ref.shape, generator=g)
    (ref * gy).sum().backward()
    wd = conv.weight.detach().to(dev())
    w2d, wt = ops.pack_conv1x1_weight(wd)
    xd = nhwc(x.detach())
    out = ops.upsample2x_add_fwd(ops.conv1x1_fwd(xd, w2d), conv.bias.detach().to(dev()), nhwc(skip))
    close(nchw(out), ref.detach(), tol=2e-6, name="bilinear up-conv fwd")
    dt = ops.upsample2x_bwd(nhwc(gy))
    close(nchw(ops.conv1x1_bwd_data(dt, wt)), x.grad, tol=2e-6, name="bilinear up-conv dgrad")
    close(ops.conv1x1_bwd_weight(xd, dt).cpu(), conv.weight.grad, tol=2e-6, name="conv1x1 wgrad")
    close(ops.channel_sum(nhwc(gy)).cpu(), conv.bias.grad, tol=2e-6, name="conv1x1 bias grad = sum of fine gradient")


def test_masked_l1_known_answers(g4):
    from resdepth_amd import masked_l1_loss
    t = lambda k: torch.from_numpy(g4["l1/" + k].copy())
    yp = t("yp").to(dev()).requires_grad_(True)
    loss = masked_l1_loss(yp, t("yt"), t("mask"), t("mean"), t("std"))
    loss.backward()
    np.testing.assert_allclose(float(loss), float(g4["l1/loss"]), rtol=2e-6)
    np.testing.assert_allclose(yp.grad.cpu().numpy(), g4["l1/gyp"], rtol=2e-6, atol=0)
    yp2 = t("yp").to(dev()).requires_grad_(True)
    loss0 = masked_l1_loss(yp2, t("yt"), torch.zeros_like(t("mask")), t("mean"), t("std"))
    loss0.backward()
    assert np.isnan(float(loss0)) and np.isnan(float(g4["l1/loss_allmasked"]))
    np.testing.assert_array_equal(yp2.grad.cpu().numpy(), g4["l1/gyp_allmasked"])


def test_masked_l1_random_vs_oracle():
    from oracle import unet_oracle as O
    from resdepth_amd import masked_l1_loss
    b = O.synthetic_batch(5, 1, 64, seed=11)
    g = torch.Generator().manual_seed(5)
    yp = (b["target"] + 0.2 * torch.randn(b["target"].shape, generator=g)).requires_grad_(True)
    std = torch.rand(5, generator=g) * 3 + 0.5
    ref = O.masked_l1_loss(yp, b["target"], b["loss_mask"], b["dsm_mean"], std)
    ref.backward()
    ypd = yp.detach().to(dev()).requires_grad_(True)
    loss = masked_l1_loss(ypd, b["target"], b["loss_mask"], b["dsm_mean"], std)
    (loss * 2.5).backward()
    np.testing.assert_allclose(float(loss), float(ref), rtol=2e-6)
    np.testing.assert_allclose(ypd.grad.cpu().numpy(), 2.5 * yp.grad.numpy(), rtol=3e-6, atol=0)


@pytest.mark.parametrize("momentum,dampening,nesterov", [(0.0, 0.0, False), (0.9, 0.0, False), (0.8, 0.1, False), (0.9, 0.0, True)])
def test_sgd_against_torch_optim_sgd(momentum, dampening, nesterov):
    """FusedSGD (lib/utils.py:332-334 configures SGD(lr, weight_decay); momentum variants are torch's) against
    torch.optim.SGD on the host: flat single-launch path (two tensors tiling one buffer) and the per-tensor path, 4 steps;
    state_dict round trip into torch.optim.SGD."""
    from resdepth_amd import FusedSGD
    g = torch.Generator().manual_seed(5)
    flat = torch.randn(300, generator=g)
    grads = torch.randn(4, 300, generator=g)
    kw = dict(lr=0.05, weight_decay=1e-2, momentum=momentum, dampening=dampening, nesterov=nesterov)
    rp = [torch.nn.Parameter(flat[:200].clone().view(10, 20)), torch.nn.Parameter(flat[200:].clone())]
    ref = torch.optim.SGD(rp, **kw)
    for mode in ("flat", "per_tensor"):
        buf, gbuf = flat.clone().to(dev()), torch.zeros(300, device=dev())
        if mode == "flat":
            ps = [torch.nn.Parameter(buf[:200].view(10, 20)), torch.nn.Parameter(buf[200:])]
        else:
            ps = [torch.nn.Parameter(buf[:200].clone().view(10, 20)), torch.nn.Parameter(buf[200:].clone())]
        opt = FusedSGD(ps, **kw)
        rp[0].data.copy_(flat[:200].view(10, 20)); rp[1].data.copy_(flat[200:])
        ref = torch.optim.SGD(rp, **kw)
        for it in range(4):
            gbuf.copy_(grads[it])
            ps[0].grad, ps[1].grad = (gbuf[:200].view(10, 20), gbuf[200:]) if mode == "flat" else \
                (gbuf[:200].clone().view(10, 20), gbuf[200:].clone())
            rp[0].grad, rp[1].grad = grads[it][:200].clone().view(10, 20), grads[it][200:].clone()
            opt.step(); ref.step()
            for a, b in zip(ps, rp):
                np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().numpy(), rtol=3e-6, atol=1e-7, err_msg=f"{mode} step {it}")
        if momentum:
            assert (mode == "flat") == bool(opt._flat_state)
            sd = opt.state_dict()
            sd["state"] = {k: {kk: (vv.cpu() if torch.is_tensor(vv) else vv) for kk, vv in v.items()} for k, v in sd["state"].items()}
            other = torch.optim.SGD([torch.nn.Parameter(torch.zeros(10, 20)), torch.nn.Parameter(torch.zeros(100))], **kw)
            other.load_state_dict(sd)
            for a, b in zip(other.param_groups[0]["params"], rp):
                np.testing.assert_allclose(other.state[a]["momentum_buffer"].numpy(), ref.state[b]["momentum_buffer"].numpy(),
                                           rtol=3e-6, atol=1e-7)


def test_adam_known_answers(g4):
    from resdepth_amd import FusedAdam
    p = torch.nn.Parameter(torch.from_numpy(g4["adam/p0"].copy()).to(dev()))
    opt = FusedAdam([p], lr=2e-4, weight_decay=1e-5)
    gs = torch.from_numpy(g4["adam/g"].copy()).to(dev())
    p.grad = gs[0].clone()
    opt.step()
    np.testing.assert_allclose(p.detach().cpu().numpy(), g4["adam/p1"], rtol=2e-6, atol=1e-8)
    opt.state[p]["step"] = torch.tensor(999.0)
    p.grad = gs[1].clone()
    opt.step()
    np.testing.assert_allclose(p.detach().cpu().numpy(), g4["adam/p1000"], rtol=2e-6, atol=1e-8)
    np.testing.assert_allclose(opt.state[p]["exp_avg"].cpu().numpy(), g4["adam/m1000"], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(opt.state[p]["exp_avg_sq"].cpu().numpy(), g4["adam/v1000"], rtol=2e-6, atol=1e-12)
    # state_dict interchangeable with torch.optim.Adam
    ref = torch.optim.Adam([torch.nn.Parameter(torch.zeros(257))], lr=2e-4, weight_decay=1e-5)
    sd = opt.state_dict()
    sd["state"] = {k: {kk: (vv.cpu() if torch.is_tensor(vv) else vv) for kk, vv in v.items()} for k, v in sd["state"].items()}
    ref.load_state_dict(sd)
    assert float(ref.state[ref.param_groups[0]["params"][0]]["step"]) == 1000.0


def test_layout_roundtrip():
    from resdepth_amd import ops
    x = torch.randn(3, 5, 8, 4).to(dev())
    y = ops.nchw_to_nhwc(x)
    assert torch.equal(y, x.permute(0, 2, 3, 1).contiguous())
    assert torch.equal(ops.nhwc_to_nchw(y), x)


@pytest.mark.parametrize("n,h,w,cin,cout,mode,kind", [
    (2, 32, 32, 64, 128, 1, "conv"), (8, 64, 64, 64, 128, 1, "conv"), (16, 64, 64, 128, 256, 2, "conv"), (2, 16, 16, 20, 36, 1, "conv"),
    (3, 24, 40, 32, 64, 1, "conv"), (2, 48, 80, 64, 128, 2, "conv"), (32, 8, 8, 512, 512, 1, "conv"), (33, 8, 8, 256, 64, 2, "conv"), (4, 16, 16, 128, 128, 1, "convt"), (2, 8, 8, 256, 256, 1, "convt"),
    (2, 6, 10, 8, 12, 1, "convt"), (4, 32, 32, 128, 128, 1, "convt"), (2, 48, 48, 256, 128, 1, "convt"), (1, 80, 80, 64, 192, 1, "convt"),
    (2, 64, 64, 64, 1, 1, "last"), (1, 24, 40, 32, 1, 1, "last"), (2, 16, 16, 8, 1, 1, "last")])
def test_bn_backward_statistics_from_the_data_gradient_epilogues(n, h, w, cin, cout, mode, kind):
    """rd_*_bwd_data_bnstats: the data gradient is bit-identical to the plain entry point, and the per-tile partial rows,
    summed by rd_bn_bwd_stats_finalize, equal the sums of the stand-alone reduction pass rd_bn_act_bwd_reduce over
    (z, g) -- halo / generic / transposed / last-conv tile kernels, both modes, LeakyReLU slope."""
    from resdepth_amd import ops
    g = torch.Generator().manual_seed(n * 1000 + h * 10 + cin)
    C = cin                                   # the BN block sits on the convolution's INPUT side
    if kind == "conv":
        dz = torch.randn(n, h, w, cout, generator=g).to(dev())
        _, wd = ops.pack_conv3x3_weight((torch.randn(cout, cin, 3, 3, generator=g) * 0.05).to(dev()))
        run = lambda **k: ops.conv3x3_bwd_data(dz, wd, **k)
    elif kind == "convt":
        dz = torch.randn(n, 2 * h, 2 * w, cout, generator=g).to(dev())
        _, wd = ops.pack_convt2x2_weight((torch.randn(cin, cout, 2, 2, generator=g) * 0.05).to(dev()))
        run = lambda **k: ops.convt2x2_bwd_data(dz, wd, **k)
    else:
        dz = torch.randn(n, 1, h, w, generator=g).to(dev())
        wt = (torch.randn(1, cin, 3, 3, generator=g) * 0.1).to(dev())
        run = lambda **k: ops.conv3x3_last_bwd_data(dz, wt, C, **k)
    z = torch.randn(n, h, w, C, generator=g).to(dev())
    mean, invstd = (torch.randn(C, generator=g) * 0.1).to(dev()), (torch.rand(C, generator=g) + 0.5).to(dev())
    gamma, beta = torch.randn(C, generator=g).to(dev()), (torch.randn(C, generator=g) * 0.3).to(dev())
    gout, part = run(bn=ops.BnHook(z, mean, invstd, gamma, beta, 0.01, None, mode))
    assert torch.equal(gout, run())
    if kind == "last" and C not in (16, 32, 64):
        assert part[1] == 0                   # no tile kernel for this channel count: the caller runs the reduction pass
        return
    assert part[1] > 0
    dg, db, de = (torch.empty(C, device=dev()) for _ in range(3))
    sums = ops.bn_bwd_stats_finalize([part], C, dg, db, de)
    ref = ops.bn_act_bwd_reduce(z, mean, invstd, gamma, beta, 0.01, gout, None, None)
    if mode == 2:
        ref[2 * C:3 * C] = 0                  # sum g belongs to the un-pooled operand only
    scale = ref.abs().view(4, C).amax(1, keepdim=True).expand(4, C).reshape(-1) + 1e-30
    assert float(((sums - ref).abs() / scale).max()) <= 1e-5
    assert torch.allclose(db.double(), ref[:C], rtol=1e-5, atol=1e-5 * float(scale[0]))
    assert torch.allclose(dg.double(), ref[C:2 * C], rtol=1e-5, atol=1e-5 * float(scale[C]))
    assert torch.allclose(de.double(), ref[2 * C:3 * C], rtol=1e-5, atol=1e-5 * float(scale[2 * C]))
    # two producers (an encoder block: un-pooled + pooled operand) add up
    both = ops.bn_bwd_stats_finalize([part, part], C)
    assert torch.allclose(both, 2 * sums, rtol=1e-12, atol=0)


def test_errors_are_reported():
    from resdepth_amd import ops
    with pytest.raises(RuntimeError, match="multiple of 4"):
        ops.conv3x3_fwd(torch.zeros(1, 4, 4, 3, device=dev()), torch.zeros(8, 9, 3, device=dev()))
    with pytest.raises(RuntimeError, match="multiple of 16"):        # the pooling epilogue lives in the patch kernels
        ops.conv3x3_fwd_act(torch.zeros(1, 12, 20, 4, device=dev()), ops.pack_conv3x3_weight_folded(
            torch.zeros(8, 4, 3, 3, device=dev()), torch.ones(8, device=dev())), torch.zeros(8, device=dev()), 0.0, pool=True)


@pytest.mark.parametrize("n,h,w,cin,cout", [(2, 8, 16, 8, 24), (8, 64, 64, 32, 128), (3, 32, 32, 20, 36), (2, 16, 16, 128, 256)])
def test_conv_epilogue_statistics_match_separate_pass(n, h, w, cin, cout):
    """BN batch statistics fused into the conv epilogues == the stand-alone statistics kernel == torch."""
    from resdepth_amd import ops
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
    wf, _ = ops.pack_conv3x3_weight(wt.to(dev()))
    z, sums = ops.conv3x3_fwd_stats(nhwc(x), wf)
    assert torch.equal(z, ops.conv3x3_fwd(nhwc(x), wf))
    ref = ops.bn_stats_partial(z)
    close(sums.cpu(), ref.cpu(), tol=1e-6, name="fused vs separate")
    zc = nchw(z).double()
    close(sums[:cout].cpu(), zc.sum((0, 2, 3)), tol=1e-6, name="sum")
    close(sums[cout:].cpu(), (zc * zc).sum((0, 2, 3)), tol=1e-6, name="sumsq")
    x3 = torch.randn(n, 3, h, w, generator=g)
    w3 = torch.randn(cout, 3, 3, 3, generator=g) / 5
    z3, s3 = ops.conv3x3_first_fwd_stats(x3.to(dev()), w3.to(dev()))
    assert torch.equal(z3, ops.conv3x3_first_fwd(x3.to(dev()), w3.to(dev())))
    close(s3.cpu(), ops.bn_stats_partial(z3).cpu(), tol=1e-6, name="first conv fused vs separate")
    # one-launch finalisation (conv + reduce/finalize): identical statistics, running buffers and counter as the two-step path
    for fused, args in ((ops.conv3x3_fwd_bn, (nhwc(x), wf)), (ops.conv3x3_first_fwd_bn, (x3.to(dev()), w3.to(dev())))):
        s_ref = sums if fused is ops.conv3x3_fwd_bn else s3
        rm = torch.linspace(-1, 1, cout, device=dev()); rv = torch.linspace(0.5, 2, cout, device=dev())
        nbt = torch.tensor(7, device=dev())
        rm2, rv2, nbt2 = rm.clone(), rv.clone(), nbt.clone()
        zf, mean, invstd = fused(*args, rm, rv, nbt)
        m_ref, i_ref = ops.bn_stats_finalize(s_ref, n * h * w, rm2, rv2, nbt2)
        assert torch.equal(zf, z if fused is ops.conv3x3_fwd_bn else z3)
        close(mean.cpu(), m_ref.cpu(), tol=1e-6, name="fused mean")
        close(invstd.cpu(), i_ref.cpu(), tol=1e-6, name="fused invstd")
        close(rm.cpu(), rm2.cpu(), tol=1e-6, name="running mean")
        close(rv.cpu(), rv2.cpu(), tol=1e-6, name="running var")
        assert int(nbt) == 8 and int(nbt2) == 8


DEV = "cuda:0"


def test_fused_sgd_mixed_momentum_state_matches_torch():
    """Some parameters already have a momentum buffer, others do not (a step taken while their gradient was None): torch
    starts the missing buffers as clone(grad).  With dampening != 0 a single `first` flag for the flat group got that wrong."""
    from resdepth_amd import FusedSGD
    g = torch.Generator().manual_seed(5)
    shapes = [(8, 4), (16,), (3, 3, 2)]
    flat = torch.randn(sum(int(np.prod(s)) for s in shapes), generator=g).to(DEV)
    init = flat.clone()
    kw = dict(lr=0.1, momentum=0.9, dampening=0.3, weight_decay=1e-2)

    def make(base):
        ps, o = [], 0
        for s in shapes:
            n = int(np.prod(s))
            ps.append(torch.nn.Parameter(base[o:o + n].view(s)))
            o += n
        return ps

    ours, ref = make(flat), [torch.nn.Parameter(p.detach().cpu().clone()) for p in make(init.clone())]
    oo, ro = FusedSGD(ours, **kw), torch.optim.SGD(ref, **kw)
    gflat = torch.zeros_like(flat)
    for step in range(3):
        grads = [torch.randn(s, generator=g) for s in shapes]
        o = 0
        for i, (p, r, gr) in enumerate(zip(ours, ref, grads)):
            n = gr.numel()
            if step == 0 and i == 1:             # step 0: the middle parameter has no gradient -> no buffer yet
                p.grad, r.grad = None, None
            else:
                gflat[o:o + n] = gr.flatten().to(DEV)
                p.grad, r.grad = gflat[o:o + n].view(gr.shape), gr.clone()
            o += n
        oo.step()
        ro.step()
    for p, r in zip(ours, ref):
        np.testing.assert_allclose(p.detach().cpu().numpy(), r.detach().numpy(), rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("n,cin,cout", [(32, 512, 512), (33, 128, 192), (16, 256, 64)])
def test_split_k_patch_kernel_for_8x8_images(n, cin, cout):
    """8 x 8 images (the cfg-S bottleneck): two images per patch, several K ranges per output tile, the last block adds the
    partials in K order (rd_set_splitk_workspace): same result as the generic kernel up to summation order, bit-identical from
    run to run, odd image counts, and the fused epilogues (BN forward statistics, BN-backward hook) see the finished tile."""
    from resdepth_amd import ops, _lib
    _lib.ensure_splitk_workspace(dev())
    g = torch.Generator().manual_seed(n + cin)
    x = torch.randn(n, cin, 8, 8, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
    wf, wd = ops.pack_conv3x3_weight(wt.to(dev()))
    ref = F.conv2d(x.double(), wt.double(), None, 1, 1)
    run = lambda: ops.conv3x3_fwd_stats(nhwc(x), wf)
    try:
        _lib.tune_set("nt_splitk", 0)
        z0, s0 = run()
        _lib.tune_set("nt_splitk", -1)
        z1, s1 = run()
        z2, s2 = run()
    finally:
        _lib.tune_set("nt_splitk", -1)
    assert torch.equal(z1, z2) and torch.equal(s1, s2)                       # run-to-run bit identity
    close(nchw(z1), ref.float(), name="split-K patch kernel vs fp64")
    close(nchw(z0), ref.float(), name="generic kernel vs fp64")
    if cin > 128:      # one K range (128 channels) adds in the generic kernel's order: same bits; more ranges: a different order
        assert not torch.equal(z0, z1), "the split-K patch kernel did not engage (same bits as the generic kernel)"
    c = z1.shape[-1]
    zz = z1.double().reshape(-1, c)
    np.testing.assert_allclose(s1[:c].cpu().numpy(), zz.sum(0).cpu().numpy(), rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(s1[c:].cpu().numpy(), (zz * zz).sum(0).cpu().numpy(), rtol=1e-5, atol=1e-4)
    # data gradient with the BN-backward hook through the same kernel
    dz = torch.randn(n, 8, 8, cout, generator=g).to(dev())
    zb = torch.randn(n, 8, 8, cin, generator=g).to(dev())
    mean, invstd = (torch.randn(cin, generator=g) * 0.1).to(dev()), (torch.rand(cin, generator=g) + 0.5).to(dev())
    gamma, beta = torch.randn(cin, generator=g).to(dev()), (torch.randn(cin, generator=g) * 0.3).to(dev())
    gout, part = ops.conv3x3_bwd_data(dz, wd, bn=ops.BnHook(zb, mean, invstd, gamma, beta, 0.01, None, 1))
    assert torch.equal(gout, ops.conv3x3_bwd_data(dz, wd)) and part[1] > 0
    sums = ops.bn_bwd_stats_finalize([part], cin)
    want = ops.bn_act_bwd_reduce(zb, mean, invstd, gamma, beta, 0.01, gout, None, None)
    scale = want.abs().view(4, cin).amax(1, keepdim=True).expand(4, cin).reshape(-1) + 1e-30
    assert float(((sums - want).abs() / scale).max()) <= 1e-5


def test_8x8_patch_kernel_does_not_depend_on_the_batch_size():
    """The 8 x 8 patch kernel is chosen by the layer shape alone and adds its K ranges in one fixed order, whether one block
    runs them all (large batches: enough tiles without a split) or one block each with the fix-up (small batches), and whether
    or not the stream has the split-K scratch: an image's result is the same bits in a batch of 3, 32 or 160 -- what tiled
    inference relies on (tests/test_blend_gpu.py: ragged batches == one full batch)."""
    from resdepth_amd import ops, _lib
    _lib.ensure_splitk_workspace(dev())
    g = torch.Generator().manual_seed(5)
    x = torch.randn(160, 512, 8, 8, generator=g)
    wt = torch.randn(512, 512, 3, 3, generator=g) / (3 * 512 ** 0.5)
    wf, wd = ops.pack_conv3x3_weight(wt.to(dev()))
    xs = nhwc(x)
    big = ops.conv3x3_fwd(xs, wf)                                 # 80 x 8 tiles x 4 ranges > 1024 blocks: unsplit
    for n in (3, 32):
        assert torch.equal(ops.conv3x3_fwd(xs[:n].contiguous(), wf), big[:n]), n
        assert torch.equal(ops.conv3x3_fwd(xs[160 - n:].contiguous(), wf), big[160 - n:]), n
    side = torch.cuda.Stream()                                    # a stream without the scratch: unsplit at any size
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        small = ops.conv3x3_fwd(xs[:3].contiguous(), wf)
    side.synchronize()
    assert torch.equal(small, big[:3])
    close(nchw(big[:8]), F.conv2d(x[:8].double(), wt.double(), None, 1, 1).float(), name="8x8 patch kernel vs fp64")


@pytest.fixture(params=["vector", "matrix"])
def first_wgrad_pipe(request):
    """The fused first weight gradient on the exact-f32 matrix pipe (the default since r05) or on the vector ALU (segment kernel:
    knob edge_conv without bit 128)."""
    from resdepth_amd import _lib
    _lib.load()
    _lib.tune_set("edge_conv", -1 if request.param == "matrix" else 63)
    yield request.param
    _lib.tune_set("edge_conv", -1)


@pytest.mark.parametrize("n,h,w,cin,cout,slope,training", [(2, 64, 64, 3, 64, 0.0, True), (3, 32, 96, 1, 32, 0.01, True),
                                                           (1, 48, 34, 2, 128, 0.01, False), (2, 18, 30, 3, 64, 0.0, True),
                                                           (2, 32, 64, 2, 64, 0.01, False)])
def test_first_conv_weight_gradient_with_the_bn_backward_evaluated_on_the_fly(n, h, w, cin, cout, slope, training, first_wgrad_pipe):
    """rd_conv3x3_first_bwd_weight_bn == rd_conv3x3_first_bwd_weight(x, rd_bn_act_bwd_apply(...)): the first block's dz (BN +
    activation + un-pool + skip add, lib/UNet.py:44-47,159-161 differentiated) has the weight gradient as its only reader and
    is evaluated inside it; tiles that are not multiples of 16 x 32, PReLU slope on the device, eval-mode form.  Both
    implementations: the segment kernel (vector ALU) and the r05 matrix-pipe kernel (v_mfma_f32_32x32x2_f32, K = pixels)."""
    from resdepth_amd import ops
    g = torch.Generator().manual_seed(h * 7 + cout)
    x = torch.randn(n, cin, h, w, generator=g).to(dev())
    z = torch.randn(n, h, w, cout, generator=g).to(dev())
    g_full = torch.randn(n, h, w, cout, generator=g).to(dev())
    g_pool = torch.randn(n, h // 2, w // 2, cout, generator=g).to(dev())
    idx = torch.randint(0, 4, (n, h // 2, w // 2, cout), generator=g, dtype=torch.uint8).to(dev())
    mean, invstd = (torch.randn(cout, generator=g) * 0.2).to(dev()), (torch.rand(cout, generator=g) + 0.5).to(dev())
    gamma, beta = torch.randn(cout, generator=g).to(dev()), (torch.randn(cout, generator=g) * 0.3).to(dev())
    sums = ops.bn_act_bwd_reduce(z, mean, invstd, gamma, beta, slope, g_full, g_pool, idx)
    count = n * h * w
    assert ops.conv3x3_first_bwd_weight_bn_available(x, cout)
    for sdev in (None, torch.full((1,), 0.2, device=dev())):
        for gf, gp, ix in ((g_full, g_pool, idx), (None, g_pool, idx)):
            dz = ops.bn_act_bwd_apply(z, mean, invstd, gamma, beta, slope, gf, gp, ix, sums, count, training, slope_dev=sdev)
            want = ops.conv3x3_first_bwd_weight(x, dz)
            got = ops.conv3x3_first_bwd_weight_bn(x, z, mean, invstd, gamma, beta, slope, gf, gp, ix, sums, count, training,
                                                  slope_dev=sdev)
            scale = float(want.abs().max())
            assert float((got - want).abs().max()) <= 2e-6 * scale, (float((got - want).abs().max()), scale)
    # the full-resolution operand as the last convolution's data gradient of a 1-channel dout, evaluated inside (LastConvGrad)
    if cout <= 64:
        dout = torch.randn(n, 1, h, w, generator=g).to(dev())
        wl = (torch.randn(1, cout, 3, 3, generator=g) / 3).to(dev())
        gl = ops.conv3x3_last_bwd_data(dout, wl, cout)
        sums_l = ops.bn_act_bwd_reduce(z, mean, invstd, gamma, beta, slope, gl, g_pool, idx)
        dz = ops.bn_act_bwd_apply(z, mean, invstd, gamma, beta, slope, gl, g_pool, idx, sums_l, count, training)
        want = ops.conv3x3_first_bwd_weight(x, dz)
        got = ops.conv3x3_first_bwd_weight_bn(x, z, mean, invstd, gamma, beta, slope, ops.LastConvGrad(dout, wl, cout), g_pool, idx,
                                              sums_l, count, training)
        scale = float(want.abs().max())
        assert float((got - want).abs().max()) <= 2e-6 * scale, (float((got - want).abs().max()), scale)
        # ... and the statistics-only form of the producer: same partial rows, no tensor
        hook = ops.BnHook(z, mean, invstd, gamma, beta, slope, None, 1)
        ds1, part1 = ops.conv3x3_last_bwd_data(dout, wl, cout, bn=hook)
        ds0, part0 = ops.conv3x3_last_bwd_data(dout, wl, cout, bn=hook, write=False)
        assert ds0 is None and part0[1] == part1[1] > 0 and torch.equal(part0[0][:part0[1] * 4 * cout], part1[0][:part1[1] * 4 * cout])
    with pytest.raises(RuntimeError, match="shape not handled"):
        ops.conv3x3_first_bwd_weight_bn(torch.randn(1, 4, 16, 32, device=dev()), z[:1, :16, :32].contiguous(), mean, invstd, gamma,
                                        beta, slope, None, g_pool[:1, :8, :16].contiguous(), idx[:1, :8, :16].contiguous(), sums, 512)


@pytest.mark.parametrize("n,h,w,cin,c0", [(2, 64, 64, 128, 64), (3, 32, 96, 64, 32), (1, 36, 20, 32, 16), (2, 16, 16, 256, 64), (2, 32, 32, 16, 16)])
def test_tail_data_gradient_of_the_last_up_convolution_from_the_output_gradient(n, h, w, cin, c0):
    """rd_tail_compose + rd_convt_last_bwd_data: the last up-convolution's input gradient as a 16-tap stride-2 stencil on dout
    == ConvTranspose2d's data gradient of the last convolution's data gradient (lib/UNet.py:21,218-227 differentiated), and
    the same BN-backward statistics as the hook of the two-kernel route; coarse grids that are not multiples of 16 x 32."""
    from resdepth_amd import ops
    g = torch.Generator().manual_seed(h + cin)
    dout = torch.randn(n, 1, h, w, generator=g)
    wt = torch.randn(cin, c0, 2, 2, generator=g) / (cin ** 0.5)
    wl = torch.randn(1, c0, 3, 3, generator=g) / 3
    # torch: g = conv_last^T(dout) [N, C0, H, W]; dprev = conv2d(g, wt as [Cin, C0, 2, 2], stride 2) = the convT's data gradient
    g4 = F.conv_transpose2d(dout.double(), wl.double(), padding=1)
    want = F.conv2d(g4, wt.double(), stride=2)
    m, v = ops.tail_compose(wt.to(dev()), wl.to(dev()))
    m_ref = torch.einsum("ioab,ot->iabt", wt.double(), wl.double()[0].reshape(c0, 9)).reshape(cin, 4, 9)
    close(m.cpu(), m_ref.float(), name="M")
    assert ops.tail_available(cin, c0)
    got = ops.convt_last_bwd_data(dout.to(dev()), v)
    close(nchw(got), want.float(), name="composed data gradient")
    # the statistics hook against the two-kernel route's
    zb = torch.randn(n, h // 2, w // 2, cin, generator=g).to(dev())
    mean, invstd = (torch.randn(cin, generator=g) * 0.1).to(dev()), (torch.rand(cin, generator=g) + 0.5).to(dev())
    gamma, beta = torch.randn(cin, generator=g).to(dev()), (torch.randn(cin, generator=g) * 0.3).to(dev())
    hook = ops.BnHook(zb, mean, invstd, gamma, beta, 0.01, None, 1)
    got2, part = ops.convt_last_bwd_data(dout.to(dev()), v, bn=hook)
    assert torch.equal(got2, got) and part[1] > 0
    sums = ops.bn_bwd_stats_finalize([part], cin)
    want_sums = ops.bn_act_bwd_reduce(zb, mean, invstd, gamma, beta, 0.01, got, None, None)
    scale = want_sums.abs().view(4, cin).amax(1, keepdim=True).expand(4, cin).reshape(-1) + 1e-30
    assert float(((sums - want_sums).abs() / scale).max()) <= 1e-5


@pytest.mark.parametrize("n,h,w,cin,c0", [(2, 64, 64, 128, 64), (3, 32, 96, 64, 32), (1, 36, 20, 32, 16), (2, 16, 16, 256, 64), (2, 32, 32, 16, 16)])
def test_tail_weight_gradient_of_the_last_up_convolution_from_the_output_gradient(n, h, w, cin, c0):
    """rd_convt_last_bwd_weight: 16 correlations of the up-convolution's input with dout per input channel, contracted with the
    last convolution's weight == ConvTranspose2d's weight gradient against the last convolution's data gradient."""
    from resdepth_amd import ops
    g = torch.Generator().manual_seed(h + cin + 1)
    dout = torch.randn(n, 1, h, w, generator=g)
    x = torch.randn(n, cin, h // 2, w // 2, generator=g)
    wl = torch.randn(1, c0, 3, 3, generator=g) / 3
    g4 = F.conv_transpose2d(dout.double(), wl.double(), padding=1)                  # [N, C0, H, W]
    wt = torch.zeros(cin, c0, 2, 2, dtype=torch.float64, requires_grad=True)
    (F.conv_transpose2d(x.double(), wt, stride=2) * g4).sum().backward()
    got = ops.convt_last_bwd_weight(nhwc(x), dout.to(dev()), wl.to(dev()))
    close(got.cpu(), wt.grad.float(), name="composed weight gradient")
    want2 = ops.convt2x2_bwd_weight(nhwc(x), ops.conv3x3_last_bwd_data(dout.to(dev()), wl.to(dev()), c0))
    close(got.cpu(), want2.cpu(), name="vs the two-kernel route")


@pytest.fixture(params=["vector", "matrix"])
def head_pipe(request):
    """The fused head of the backward (rd_conv3x3_last_bwd_tail_fused) on the vector ALU or on the exact-f32 matrix pipe (knob
    edge_conv bit 256)."""
    from resdepth_amd import _lib
    _lib.load()
    _lib.tune_set("edge_conv", HEAD_KNOB[request.param])
    yield request.param
    _lib.tune_set("edge_conv", -1)


HEAD_KNOB = {"vector": 255, "matrix": 255 | 256}


@pytest.mark.parametrize("n,h,w,cin,c0,slope,res", [(2, 64, 64, 128, 64, 0.0, True), (3, 32, 96, 64, 32, 0.01, False),
                                                    (1, 36, 20, 32, 16, 0.01, True), (2, 32, 32, 16, 16, 0.0, True),
                                                    (2, 18, 44, 64, 64, 0.01, True)])
def test_tail_forward_and_last_weight_gradient_without_the_up_convolution_output(n, h, w, cin, c0, slope, res, head_pipe):
    """rd_conv3x3_last_fwd_tail / rd_conv3x3_last_bwd_weight_tail: the last convolution applied to
    s = ConvTranspose2d(x_coarse) + act(BN(z)) (lib/UNet.py:218-227) without s ever being a tensor -- forward from z (BN +
    activation on load), T = x_coarse . V and the bias stencil; its weight / bias gradient from z, dout and the correlations C16.
    Against torch in fp64 on the materialised s, borders included."""
    from resdepth_amd import ops
    g = torch.Generator().manual_seed(h * 3 + cin)
    xc = torch.randn(n, cin, h // 2, w // 2, generator=g)
    z = torch.randn(n, c0, h, w, generator=g)
    wt = torch.randn(cin, c0, 2, 2, generator=g) / (cin ** 0.5)
    bt = torch.randn(c0, generator=g) * 0.2
    wl = torch.randn(1, c0, 3, 3, generator=g) / 3
    bl = torch.randn(1, generator=g)
    xin = torch.randn(n, 3, h, w, generator=g)
    mean, invstd = torch.randn(c0, generator=g) * 0.2, torch.rand(c0, generator=g) + 0.5
    gamma, beta = torch.randn(c0, generator=g), torch.randn(c0, generator=g) * 0.3
    # reference in fp64
    y = (z.double() - mean.double().view(1, -1, 1, 1)) * (invstd.double() * gamma.double()).view(1, -1, 1, 1) + beta.double().view(1, -1, 1, 1)
    a0 = torch.where(y > 0, y, y * slope)
    s = F.conv_transpose2d(xc.double(), wt.double(), bt.double(), stride=2) + a0
    wl64 = wl.double().requires_grad_(True)
    bl64 = bl.double().requires_grad_(True)
    out_ref = F.conv2d(s, wl64, bl64, padding=1) + (xin.double()[:, 0:1] if res else 0)
    dout = torch.randn(n, 1, h, w, generator=g)
    (out_ref * dout.double()).sum().backward()
    # HIP
    D = dev()
    skip = {"z": nhwc(z), "mean": mean.to(D), "invstd": invstd.to(D), "gamma": gamma.to(D), "beta": beta.to(D), "slope": slope, "slope_dev": None}
    m, v, vt, b9 = ops.tail_compose(wt.to(D), wl.to(D), bt.to(D), forward=True)
    assert torch.equal(vt.view(16, cin).t().contiguous(), v)
    t16 = ops.conv1x1_fwd(nhwc(xc), ops.pack_conv1x1_weight(vt)[0])
    t16b = ops.tail_t16(nhwc(xc), v)                          # the dedicated kernel (exact-f32 matrix pipe), same contraction
    close(t16b.cpu(), torch.einsum("nchw,cd->nhwd", xc.double(), v.cpu().double()).float(), tol=3e-6, name="T = x . V")
    assert float((t16 - t16b).abs().max()) <= 3e-6 * float(t16b.abs().max())
    # ... and from the pre-BN tensor of the producing block, BN + activation on load (forward and the correlations)
    zc = torch.randn(n, cin, h // 2, w // 2, generator=g)
    mc, ic = torch.randn(cin, generator=g) * 0.2, torch.rand(cin, generator=g) + 0.5
    gc, bc = torch.randn(cin, generator=g), torch.randn(cin, generator=g) * 0.3
    yc = (zc.double() - mc.double().view(1, -1, 1, 1)) * (ic.double() * gc.double()).view(1, -1, 1, 1) + bc.double().view(1, -1, 1, 1)
    ac = torch.where(yc > 0, yc, yc * slope).float()
    desc = {"z": nhwc(zc), "mean": mc.to(D), "invstd": ic.to(D), "gamma": gc.to(D), "beta": bc.to(D), "slope": slope, "slope_dev": None}
    close(ops.tail_t16(desc, v).cpu(), ops.tail_t16(nhwc(ac), v).cpu(), tol=3e-6, name="T from the pre-BN tensor")
    dprobe = torch.randn(n, 1, h, w, generator=g).to(D)
    close(ops.convt_last_bwd_weight(desc, dprobe, wl.to(D)).cpu(), ops.convt_last_bwd_weight(nhwc(ac), dprobe, wl.to(D)).cpu(),
          tol=3e-6, name="correlations from the pre-BN tensor")
    out = ops.conv3x3_last_fwd_tail(skip, t16b, b9, wl.to(D), bl.to(D), xin.to(D) if res else None)
    close(out.cpu(), out_ref.detach().float(), tol=3e-6, name="tail forward")
    c16 = torch.empty(cin, 16, dtype=torch.float64, device=D)
    ops.convt_last_bwd_weight(nhwc(xc), dout.to(D), wl.to(D), c16=c16)
    dw, db = ops.conv3x3_last_bwd_weight_tail(skip, dout.to(D), c16, wt.to(D), bt.to(D))
    close(dw.cpu(), wl64.grad.float(), tol=3e-6, name="tail last-conv weight gradient")
    assert abs(float(db) - float(bl64.grad)) <= 1e-5 * max(1.0, abs(float(bl64.grad)))
    # the fused head of the backward: same weight gradient through tail_wl_finish, same statistics as the data-gradient hook
    wpart, stat = ops.conv3x3_last_bwd_tail_fused(skip, dout.to(D), wl.to(D))
    dw2, db2 = ops.tail_wl_finish(wpart, c16, wt.to(D), bt.to(D))
    close(dw2.cpu(), wl64.grad.float(), tol=3e-6, name="fused head: last-conv weight gradient")
    assert abs(float(db2) - float(bl64.grad)) <= 1e-5 * max(1.0, abs(float(bl64.grad)))
    hook = ops.BnHook(skip["z"], skip["mean"], skip["invstd"], skip["gamma"], skip["beta"], slope, None, 1)
    _, stat_ref = ops.conv3x3_last_bwd_data(dout.to(D), wl.to(D), c0, bn=hook)
    s_a, s_b = ops.bn_bwd_stats_finalize([stat], c0), ops.bn_bwd_stats_finalize([stat_ref], c0)
    scale = s_b.abs().view(4, c0).amax(1, keepdim=True).expand(4, c0).reshape(-1) + 1e-30
    assert float(((s_a - s_b).abs() / scale).max()) <= 1e-5
    # without an up-convolution bias
    out_nb = ops.conv3x3_last_fwd_tail(skip, t16, ops.tail_compose(wt.to(D), wl.to(D), None, forward=True)[3], wl.to(D), None, None)
    s_nb = F.conv_transpose2d(xc.double(), wt.double(), None, stride=2) + a0
    close(out_nb.cpu(), F.conv2d(s_nb, wl.double(), None, padding=1).float(), tol=3e-6, name="tail forward, no biases")


def test_tail_ops_on_random_shapes_against_the_two_kernel_route():
    """Seeded random coarse grids (1 x 1 up to 40 x 50, odd sizes, one image up to five) and every supported channel pair: the
    composed data gradient, weight gradient and forward of the tail against the kernels they replace."""
    from resdepth_amd import ops
    rng = np.random.RandomState(7)
    D = dev()
    for case in range(8):
        n, hc, wc = int(rng.randint(1, 6)), int(rng.randint(1, 41)), int(rng.randint(1, 51))
        cin, c0 = int(rng.choice([16, 32, 64, 128, 256])), int(rng.choice([16, 32, 64]))
        g = torch.Generator().manual_seed(100 + case)
        h, w = 2 * hc, 2 * wc
        xc = torch.randn(n, hc, wc, cin, generator=g).to(D)
        dout = torch.randn(n, 1, h, w, generator=g).to(D)
        wt = (torch.randn(cin, c0, 2, 2, generator=g) / cin ** 0.5).to(D)
        bt = (torch.randn(c0, generator=g) * 0.2).to(D)
        wl = (torch.randn(1, c0, 3, 3, generator=g) / 3).to(D)
        tag = (case, n, hc, wc, cin, c0)
        # backward: g = conv_last^T(dout) materialised, then the transposed convolution's two gradients
        g4 = ops.conv3x3_last_bwd_data(dout, wl, c0)
        _, wtd = ops.pack_convt2x2_weight(wt)
        _, v, _, b9 = ops.tail_compose(wt, wl, bt, forward=True)
        want, got = ops.convt2x2_bwd_data(g4, wtd), ops.convt_last_bwd_data(dout, v)
        assert float((got - want).abs().max()) <= 3e-6 * float(want.abs().max() + 1e-30), ("dgrad",) + tag
        want, got = ops.convt2x2_bwd_weight(xc, g4), ops.convt_last_bwd_weight(xc, dout, wl)
        assert float((got - want).abs().max()) <= 3e-6 * float(want.abs().max() + 1e-30), ("wgrad",) + tag
        # forward: s = up-convolution + identity skip of a random z (mean 0, invstd 1, gamma 1, beta 0, slope 1), last convolution
        z = torch.randn(n, h, w, c0, generator=g).to(D)
        one, zero = torch.ones(c0, device=D), torch.zeros(c0, device=D)
        skip = {"z": z, "mean": zero, "invstd": one, "gamma": one, "beta": zero, "slope": 1.0, "slope_dev": None}
        wtf, _ = ops.pack_convt2x2_weight(wt)
        s = ops.convt2x2_fwd(xc, wtf, bt, z)
        want = ops.conv3x3_last_fwd(s, wl, None, None)
        got = ops.conv3x3_last_fwd_tail(skip, ops.tail_t16(xc, v), b9, wl, None, None)
        assert float((got - want).abs().max()) <= 5e-6 * float(want.abs().max() + 1e-30), ("forward",) + tag


class _tune:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        from resdepth_amd import _lib
        self.old = {k: _lib.tune_get(k) for k in self.kv}
        for k, v in self.kv.items():
            _lib.tune_set(k, v)

    def __exit__(self, *exc):
        from resdepth_amd import _lib
        for k, v in self.old.items():
            _lib.tune_set(k, v)


@pytest.mark.parametrize("n,h,w,cin,cout", [
    (16, 64, 64, 128, 256),     # conv3_halo_split<128>
    (8, 64, 64, 128, 64),       # <64>: two wave row-bands (cross-wave statistics exchange)
    (33, 8, 8, 256, 64),        # two-image patches, odd image count: the last patch's second image is absent
    (4, 16, 16, 64, 64),        # generic NT kernel, full 64-row tiles
])
def test_register_direct_epilogues_equal_the_staged_ones(n, h, w, cin, cout):
    """r04: the patch kernels (and the generic NT kernels on full tiles) store C, the BatchNorm statistics, the BN-backward hook
    sums and the inference shift / activation / max-pool straight from the accumulator registers (rd_nt.h: nt_epilogue_direct);
    `nt_epi = 0` keeps the LDS-staged epilogue.  Same accumulators either way: C, activations and pooled values are the SAME
    BITS; the statistics are the same sums in another (fixed) order."""
    from resdepth_amd import ops
    g = torch.Generator().manual_seed(n + h + cin + cout)
    x = torch.randn(n, h, w, cin, generator=g).to(dev())
    dz = torch.randn(n, h, w, cout, generator=g).to(dev())
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).to(dev())
    wf, wd = ops.pack_conv3x3_weight(wt)
    zin = torch.randn(n, h, w, cin, generator=g).to(dev())
    hook = ops.BnHook(zin, (torch.randn(cin, generator=g) * 0.1).to(dev()), (torch.rand(cin, generator=g) + 0.5).to(dev()),
                      torch.randn(cin, generator=g).to(dev()), (torch.randn(cin, generator=g) * 0.3).to(dev()), 0.01, None, 1)
    scale = (torch.rand(cout, generator=g) + 0.5).to(dev())
    shift = (torch.randn(cout, generator=g) * 0.2).to(dev())
    wfold = ops.pack_conv3x3_weight_folded(wt, scale)
    pool = w % 16 == 0 and h % 8 == 0 and w > 8

    def run():
        z, sums = ops.conv3x3_fwd_stats(x, wf)
        dx, part = ops.conv3x3_bwd_data(dz, wd, hook)
        hs = ops.bn_bwd_stats_finalize([part], cin) if part[1] > 0 else None
        a, p = ops.conv3x3_fwd_act(x, wfold, shift, 0.01, pool=pool)
        torch.cuda.synchronize()
        return z, sums, dx, hs, a, p

    with _tune(nt_epi=0):
        z0, s0, dx0, h0, a0, p0 = run()
    z1, s1, dx1, h1, a1, p1 = run()
    assert torch.equal(z0, z1) and torch.equal(dx0, dx1) and torch.equal(a0, a1)
    assert (p0 is None) == (p1 is None) and (p0 is None or torch.equal(p0, p1))
    close(s1.cpu(), s0.cpu(), tol=1e-6, name="BN statistics")
    assert (h0 is None) == (h1 is None)
    if h0 is not None:
        sc = h0.abs().view(4, cin).amax(1, keepdim=True).expand(4, cin).reshape(-1) + 1e-30
        assert float(((h1 - h0).abs() / sc).max()) <= 1e-5


@pytest.mark.parametrize("n,cin,h,w,cout,slope,prelu", [
    (2, 3, 64, 64, 64, 0.0, False), (1, 1, 32, 96, 64, 0.01, False), (3, 2, 24, 40, 32, 0.0, False),    # ragged tiles (16 x 32)
    (2, 4, 16, 32, 128, 0.0, True), (1, 3, 18, 34, 64, 0.0, False)])
def test_first_convolution_with_bn_activation_and_pool_in_its_epilogue(n, cin, h, w, cout, slope, prelu):
    """rd_conv3x3_first_fwd_act (inference, r04) == rd_conv3x3_first_fwd followed by rd_bn_act_pool_fwd, bit for bit: the same
    fmaf chain for the convolution, the same fma + select for BN + activation, torch's window order / NaN rule for the pool."""
    from resdepth_amd import ops
    g = torch.Generator().manual_seed(n * 100 + cin * 10 + cout)
    x = torch.randn(n, cin, h, w, generator=g)
    x[0, 0, 3, 5] = float("nan")                                # a NaN must win its pooling window (and poison its 3 x 3 reach)
    x = x.to(dev())
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / 3).to(dev())
    rm, rv = (torch.randn(cout, generator=g) * 0.2).to(dev()), (torch.rand(cout, generator=g) + 0.5).to(dev())
    gamma, beta = torch.randn(cout, generator=g).to(dev()), (torch.randn(cout, generator=g) * 0.3).to(dev())
    sdev = torch.tensor([0.25], device=dev()) if prelu else None
    assert ops.conv3x3_first_fwd_act_available(x, cout)
    mean, invstd = ops.bn_eval_stats(rm, rv)
    a, p = ops.conv3x3_first_fwd_act(x, wt, mean, invstd, gamma, beta, slope, sdev, pool=True)
    z = ops.conv3x3_first_fwd(x, wt)
    a_ref, p_ref, _ = ops.bn_act_pool_fwd(z, mean, invstd, gamma, beta, slope, True, sdev, want_a=True)
    def same(u, v):
        return torch.equal(torch.nan_to_num(u, nan=12345.0), torch.nan_to_num(v, nan=12345.0)) and torch.equal(torch.isnan(u), torch.isnan(v))

    assert same(a, a_ref) and same(p, p_ref)
    assert torch.isnan(p).any()
    a2, p2 = ops.conv3x3_first_fwd_act(x, wt, mean, invstd, gamma, beta, slope, sdev, pool=False)
    assert p2 is None and same(a2, a_ref)
    # against torch on the CPU (no NaN): conv -> eval BatchNorm -> activation -> max-pool
    xc = torch.randn(n, cin, h, w, generator=g)
    zt = F.conv2d(xc, wt.cpu(), padding=1)
    yt = F.batch_norm(zt, rm.cpu(), rv.cpu(), gamma.cpu(), beta.cpu(), False, 0.1, 1e-5)
    yt = F.prelu(yt, sdev.cpu()) if prelu else F.leaky_relu(yt, slope)
    a3, p3 = ops.conv3x3_first_fwd_act(xc.to(dev()), wt, mean, invstd, gamma, beta, slope, sdev, pool=True)
    close(nchw(a3), yt, tol=2e-6, name="a")
    close(nchw(p3), F.max_pool2d(yt, 2), tol=2e-6, name="pooled")


@pytest.mark.parametrize("cin", [1, 2, 3, 4])
@pytest.mark.parametrize("cout", [32, 64])
def test_first_convolution_on_the_matrix_pipe_is_bit_identical(cin, cout):
    """r05: `encoder.0.0.0` (lib/UNet.py:159) forward on `v_mfma_f32_32x32x2_f32` -- fp32 fused multiply-adds in the k order of the
    vector kernels (opt-in, knob edge_conv = 127: measured slower, profiles/r05_notes.md section 11) -- against those kernels (the
    default): z THE SAME BITS, on full and ragged tiles; the
    BatchNorm statistics of its epilogue agree to rounding (another summation order); the inference form (eval BN + activation +
    2 x 2 max-pool in the epilogue) gives the same activation and pooled BITS, NaN / tie rule included."""
    from resdepth_amd import ops, _lib
    _lib.load()
    g = torch.Generator().manual_seed(100 * cin + cout)
    for (n, h, w) in ((3, 64, 96), (2, 48, 40), (1, 16, 8)):
        x = torch.randn(n, cin, h, w, generator=g)
        x[0, 0, 3, 5] = float("nan") if h == 48 else x[0, 0, 3, 5]
        x[-1, -1, 6:8, 6:8] = 0.25                                   # a constant patch: pool ties
        wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.3
        mean, var = torch.randn(cout, generator=g) * 0.1, torch.rand(cout, generator=g) + 0.5
        gamma, beta = torch.randn(cout, generator=g), torch.randn(cout, generator=g) * 0.2
        xd, wd = x.to(DEV), wt.to(DEV)
        invstd = (var + 1e-5).rsqrt().to(DEV)
        res = {}
        for knob in (127, -1):
            _lib.tune_set("edge_conv", knob)
            try:
                z = ops.conv3x3_first_fwd(xd, wd)
                z2, sums = ops.conv3x3_first_fwd_stats(xd, wd)
                a, p = ops.conv3x3_first_fwd_act(xd, wd, mean.to(DEV), invstd, gamma.to(DEV), beta.to(DEV), 0.01, None, pool=True)
                torch.cuda.synchronize()
            finally:
                _lib.tune_set("edge_conv", -1)
            res[knob] = (z.cpu(), z2.cpu(), sums.cpu(), a.cpu(), p.cpu())
        new, old = res[127], res[-1]
        same = lambda u, v: torch.equal(u.view(torch.int32), v.view(torch.int32))
        assert same(new[0], old[0]) and same(new[1], old[1]), (cin, cout, n, h, w)
        assert same(new[3], old[3]) and same(new[4], old[4]), (cin, cout, n, h, w)
        fin = torch.isfinite(old[2])
        assert torch.equal(fin, torch.isfinite(new[2]))
        if bool(fin.any()):
            assert float(((new[2] - old[2])[fin]).abs().max() / (old[2][fin].abs().max() + 1e-30)) <= 1e-6
        # and against torch on the CPU (what both have to be)
        ref = F.conv2d(x, wt, None, 1, 1).permute(0, 2, 3, 1)
        ok = torch.isfinite(ref)
        assert float((new[0][ok] - ref[ok]).abs().max()) <= 2e-5 * float(ref[ok].abs().max())
