"""Split-bf16 MFMA arithmetic on adversarial operands (-m gpu): every conv3x3 / convT2x2 op (forward, data gradient, weight
gradient; every tile variant of the NT kernels) against an fp64 reference AND against the exact-f32 MFMA kernels of the same
library (rd_tune_set("mfma_f32", 1)) on

  pos    all-positive activations x all-positive weights, K = 4608 (one-sided error sources cannot cancel)
  range  magnitudes 1e-30 .. 1e30 across the operand pair, log-uniform, random signs
  int24  full-mantissa integers (2^24 - 1 - 2k) x small integers (every one of the 24 operand bits matters)
  tiny   |x| ~ 1e-36: below 2^-110 the third split term is a bf16 subnormal (documented graceful degradation)
  +-Inf / NaN operands: the non-finite pattern of the output against torch's fp32 CPU convolution

Error metric: e = |out - ref64| / sum_k |a_k| |b_k| per output, in units of u = 2^-24 (one fp32 rounding); bars are
relative to the exact-f32 kernel's own error on the same data (numbers: scripts/split_numerics.py)."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))

pytestmark = pytest.mark.gpu
U = 2.0 ** -24


@pytest.fixture()
def modes():
    from resdepth_amd import _lib
    _lib.load()
    yield _lib
    _lib.tune_set("mfma_f32", 0)
    _lib.tune_set("nt_tile", -1)


def _both(lib, run):
    res = {}
    for mode in (0, 1):
        lib.tune_set("mfma_f32", mode)
        res[mode] = run()
    lib.tune_set("mfma_f32", 0)
    return res


@pytest.mark.parametrize("family,tile", [("conv", -1), ("conv", 0), ("conv", 1), ("conv", 2), ("convt", -1)])
@pytest.mark.parametrize("flavour", ["randn", "pos", "range", "int24"])
def test_split_error_within_a_small_factor_of_exact_f32(modes, family, tile, flavour):
    import split_numerics as SN
    g = torch.Generator().manual_seed(7)
    ref, den, run = (SN.conv_cases if family == "conv" else SN.convt_cases)(flavour, g)
    modes.tune_set("nt_tile", tile)
    res = _both(modes, run)
    for k in ref:
        if tile != -1 and k == "wgrad":
            continue                                   # the tile override only concerns the NT kernels
        s_max, s_rms = SN.nerr(res[0][k], ref[k], den[k])
        f_max, f_rms = SN.nerr(res[1][k], ref[k], den[k])
        # measured (r2): split/f32 rms ratio 0.3 .. 1.1 on the NT kernels (hi/lo accumulators), <= 1.6 on the weight
        # gradients; absolute floor for cases where the exact kernel happens to be almost error free
        assert s_rms <= 2.0 * f_rms + 0.5, (family, tile, flavour, k, s_rms, f_rms)
        assert s_max <= 2.5 * f_max + 4.0, (family, tile, flavour, k, s_max, f_max)
        assert torch.isfinite(res[0][k]).all()


@pytest.mark.parametrize("family", ["conv", "convt"])
def test_operands_below_2pow_minus110_degrade_gracefully(modes, family):
    """|x| ~ 1e-36 < 2^-110: the third split term is a bf16 subnormal and x is represented with an absolute error
    <= 2^-133 -- the result is off by at most that times sum |b| (a relative 2^-16-class error at this magnitude; cuDNN /
    oneDNN builds that flush denormals lose the same operands entirely).  Everything at or above 2^-110 is exact (`range`)."""
    import split_numerics as SN
    g = torch.Generator().manual_seed(7)
    ref, den, run = (SN.conv_cases if family == "conv" else SN.convt_cases)("tiny", g)
    modes.tune_set("mfma_f32", 0)
    out = run()
    for k in ("fwd", "wgrad"):           # the ops that see the tiny operand (dgrad multiplies gy x w only)
        err = (out[k].double() - ref[k]).abs()
        assert float((err / (den[k] + 1e-300)).max()) <= 2.0 ** -14, k
    s_max, _ = SN.nerr(out["dgrad"], ref["dgrad"], den["dgrad"])
    assert s_max <= 8.0


@pytest.mark.parametrize("what", ["x=+inf", "x=-inf", "x=nan", "w=+inf", "w=nan", "x=+inf,w=0"])
def test_nonfinite_operands_propagate_like_fp32(modes, what):
    """+-Inf / NaN in an activation or a weight: forward, data gradient (NT kernels with hi/lo accumulators) reproduce
    torch's fp32 pattern exactly -- Inf stays Inf with its sign, Inf * 0 and NaN give NaN, everything else stays finite.
    The weight-gradient kernels keep one accumulator: same SET of non-finite outputs, an Inf may surface as NaN."""
    from resdepth_amd import ops
    dev = "cuda:0"
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 64, 16, 16, generator=g)
    wt = torch.randn(128, 64, 3, 3, generator=g) * 0.05
    gy = torch.randn(2, 128, 16, 16, generator=g)
    val = {"+inf": float("inf"), "-inf": float("-inf"), "nan": float("nan"), "0": 0.0}
    for item in what.split(","):
        name, v = item.split("=")
        if name == "x":
            x[0, 5, 7, 9] = val[v]
            gy[1, 3, 2, 4] = val[v]
        elif v == "0":
            wt[:, 5] = 0.0                               # Inf * 0 -> NaN wherever the infinite pixel meets channel 5's taps
        else:
            wt[3, 2, 1, 1] = val[v]
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(dev)
    nchw = lambda t: t.permute(0, 3, 1, 2).contiguous().cpu()
    ref_f = F.conv2d(x, wt, None, 1, 1)
    ref_d = F.conv_transpose2d(gy, wt, None, 1, 1)
    ref_w = torch.nn.grad.conv2d_weight(x, wt.shape, gy, stride=1, padding=1)
    for tile in (-1, 0, 1, 2):
        modes.tune_set("nt_tile", tile)
        wf, wd = ops.pack_conv3x3_weight(wt.to(dev))
        for out, ref in ((nchw(ops.conv3x3_fwd(nhwc(x), wf)), ref_f), (nchw(ops.conv3x3_bwd_data(nhwc(gy), wd)), ref_d)):
            assert torch.equal(torch.isnan(out), torch.isnan(ref)), (what, tile)
            inf = torch.isinf(ref)
            assert torch.equal(torch.isinf(out), inf) and torch.equal(out[inf], ref[inf]), (what, tile)
            fin = torch.isfinite(ref)
            assert float((out[fin] - ref[fin]).abs().max()) <= 1e-4 * float(ref[fin].abs().max())
    modes.tune_set("nt_tile", -1)
    dw = ops.conv3x3_bwd_weight(nhwc(x), nhwc(gy)).cpu()
    assert torch.equal(torch.isfinite(dw), torch.isfinite(ref_w)), what
    # transposed convolution (1-tap NT kernel with the scatter epilogue, 4-tap gather data gradient)
    wtt = torch.randn(64, 64, 2, 2, generator=g) * 0.05
    if "w=" in what and "0" not in what:
        wtt[3, 2, 1, 1] = wt[3, 2, 1, 1]
    wtf, wtd = ops.pack_convt2x2_weight(wtt.to(dev))
    gy2 = torch.randn(2, 64, 32, 32, generator=g)
    gy2[1, 3, 2, 4] = gy[1, 3, 2, 4]
    for out, ref in ((nchw(ops.convt2x2_fwd(nhwc(x), wtf, None, None)), F.conv_transpose2d(x, wtt, None, 2)),
                     (nchw(ops.convt2x2_bwd_data(nhwc(gy2), wtd)), F.conv2d(gy2, wtt, None, 2))):
        assert torch.equal(torch.isnan(out), torch.isnan(ref)), what
        inf = torch.isinf(ref)
        assert torch.equal(torch.isinf(out), inf) and torch.equal(out[inf], ref[inf]), what
