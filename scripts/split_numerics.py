"""Error of the MFMA-class kernels on adversarial operands, split-bf16 vs exact-f32 (diagnosis aid behind
tests/test_split_numerics_gpu.py).  For every op and operand flavour prints

    e_max, e_rms  of  |out - ref64| / sum_k |a_k| |b_k|     in units of u = 2^-24 (one fp32 rounding)

for both arithmetic modes (rd_tune_set("mfma_f32", 0|1)), plus the non-finite pattern for the Inf / NaN flavours.

    python scripts/split_numerics.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from resdepth_amd import _lib, ops

DEV = "cuda:0"
U = 2.0 ** -24


def nhwc(t):
    # split2h mode: an operand needs its magnitude slot to take the three-product body (ops.amax_of: one rd_amax pass; a no-op
    # in the other modes) -- the engine's tensors get theirs from the producing kernel's epilogue
    return ops.amax_of(t.permute(0, 2, 3, 1).contiguous().to(DEV))


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous().cpu()


def flavour(name, shape, g, role):
    """role 'a' = activation-like operand, 'b' = weight-like operand (scaled so that sums stay finite)."""
    r = torch.randn(shape, generator=g)
    if name == "randn":
        return r if role == "a" else r * 0.05
    if name == "pos":
        return r.abs() if role == "a" else r.abs() * 0.05
    if name == "range":            # 1e-30 .. 1e30 across the pair: a in 1e-30..1e-5, b in 1e5..1e30 (log-uniform), random signs
        e = torch.rand(shape, generator=g) * 25.0
        s = torch.sign(r)
        return s * 10.0 ** (-30.0 + e) if role == "a" else s * 10.0 ** (5.0 + e)
    if name == "tiny":             # third split term of a ~1e-36 value is subnormal
        return r * 1e-36 if role == "a" else r
    if name == "int24":            # full-mantissa integers (2^24 - 1 - 2k), small-integer weights
        k = torch.randint(0, 64, shape, generator=g).float()
        s = torch.sign(r)
        return s * (16777215.0 - 2.0 * k) if role == "a" else s * torch.randint(1, 8, shape, generator=g).float() * 2.0 ** -30
    raise ValueError(name)


def nerr(out, ref, den):
    e = (out.double() - ref).abs() / (den + 1e-300)
    return float(e.max()) / U, float(e.pow(2).mean().sqrt()) / U


def conv_cases(fl, g, n=2, h=16, w=16, cin=512, cout=128):
    x = flavour(fl, (n, cin, h, w), g, "a")
    wt = flavour(fl, (cout, cin, 3, 3), g, "b")
    gy = flavour("randn" if fl == "tiny" else fl, (n, cout, h, w), g, "a")      # tiny * tiny would underflow to 0
    xd, wd_, gd = x.double(), wt.double(), gy.double()
    ref = {"fwd": F.conv2d(xd, wd_, None, 1, 1), "dgrad": F.conv_transpose2d(gd, wd_, None, 1, 1),
           "wgrad": torch.nn.grad.conv2d_weight(xd, wt.shape, gd, stride=1, padding=1)}
    den = {"fwd": F.conv2d(xd.abs(), wd_.abs(), None, 1, 1), "dgrad": F.conv_transpose2d(gd.abs(), wd_.abs(), None, 1, 1),
           "wgrad": torch.nn.grad.conv2d_weight(xd.abs(), wt.shape, gd.abs(), stride=1, padding=1)}

    def run():
        wf, wdd = ops.pack_conv3x3_weight(wt.to(DEV))
        return {"fwd": nchw(ops.conv3x3_fwd(nhwc(x), wf)), "dgrad": nchw(ops.conv3x3_bwd_data(nhwc(gy), wdd)),
                "wgrad": ops.conv3x3_bwd_weight(nhwc(x), nhwc(gy)).cpu()}
    return ref, den, run


def convt_cases(fl, g, n=2, h=16, w=16, c=512):
    x = flavour(fl, (n, c, h, w), g, "a")
    wt = flavour(fl, (c, c, 2, 2), g, "b")
    gy = flavour("randn" if fl == "tiny" else fl, (n, c, 2 * h, 2 * w), g, "a")
    xd, wd_, gd = x.double(), wt.double(), gy.double()
    ref = {"fwd": F.conv_transpose2d(xd, wd_, None, 2), "dgrad": F.conv2d(gd, wd_, None, 2),
           "wgrad": torch.nn.grad.conv2d_weight(gd, wt.shape, xd, stride=2)}
    den = {"fwd": F.conv_transpose2d(xd.abs(), wd_.abs(), None, 2), "dgrad": F.conv2d(gd.abs(), wd_.abs(), None, 2),
           "wgrad": torch.nn.grad.conv2d_weight(gd.abs(), wt.shape, xd.abs(), stride=2)}

    def run():
        wtf, wtd = ops.pack_convt2x2_weight(wt.to(DEV))
        return {"fwd": nchw(ops.convt2x2_fwd(nhwc(x), wtf, None, None)), "dgrad": nchw(ops.convt2x2_bwd_data(nhwc(gy), wtd)),
                "wgrad": ops.convt2x2_bwd_weight(nhwc(x), nhwc(gy)).cpu()}
    return ref, den, run


def main():
    _lib.load()
    print(f"{'op':28s} {'flavour':7s} | split e_max e_rms [u] | f32 e_max e_rms [u]")
    for fam, mk, tiles in (("conv3x3", conv_cases, (-1, 0, 1, 2)), ("convT2x2", convt_cases, (-1,))):
        for fl in ("randn", "pos", "range", "tiny", "int24"):
            for tile in tiles:
                g = torch.Generator().manual_seed(7)
                ref, den, run = mk(fl, g)
                _lib.tune_set("nt_tile", tile)
                res = {}
                for mode in (0, 1):
                    _lib.tune_set("mfma_f32", mode)
                    res[mode] = run()
                _lib.tune_set("mfma_f32", 0)
                for k in ref:
                    if tile != -1 and k == "wgrad":
                        continue
                    s, f = nerr(res[0][k], ref[k], den[k]), nerr(res[1][k], ref[k], den[k])
                    print(f"{fam + ' ' + k + ' tile=' + str(tile):28s} {fl:7s} | {s[0]:9.2f} {s[1]:7.3f} | {f[0]:9.2f} {f[1]:7.3f}")
    _lib.tune_set("nt_tile", -1)
    # non-finite operands: where does the output become Inf / NaN?  reference = torch fp32 CPU op
    for what in ("x=+inf", "x=nan", "w=+inf", "x=-inf"):
        g = torch.Generator().manual_seed(3)
        x = torch.randn(2, 64, 16, 16, generator=g)
        wt = torch.randn(128, 64, 3, 3, generator=g) * 0.05
        if what.startswith("x"):
            x[0, 5, 7, 9] = {"x=+inf": float("inf"), "x=-inf": float("-inf"), "x=nan": float("nan")}[what]
        else:
            wt[3, 2, 1, 1] = float("inf")
        ref = F.conv2d(x, wt, None, 1, 1)
        for mode in (0, 1):
            _lib.tune_set("mfma_f32", mode)
            wf, _ = ops.pack_conv3x3_weight(wt.to(DEV))
            out = nchw(ops.conv3x3_fwd(nhwc(x), wf))
            same_nonfinite = bool((torch.isfinite(out) == torch.isfinite(ref)).all())
            same_inf = bool((torch.isinf(out) == torch.isinf(ref)).all()) and bool(((out == ref) | ~torch.isinf(ref)).all())
            same_nan = bool((torch.isnan(out) == torch.isnan(ref)).all())
            print(f"conv3x3 fwd {what:7s} mode={'f32' if mode else 'split'}: ref inf {int(torch.isinf(ref).sum())} nan "
                  f"{int(torch.isnan(ref).sum())} | out inf {int(torch.isinf(out).sum())} nan {int(torch.isnan(out).sum())} | "
                  f"non-finite set equal {same_nonfinite}, inf equal {same_inf}, nan equal {same_nan}")
        _lib.tune_set("mfma_f32", 0)


if __name__ == "__main__":
    main()
