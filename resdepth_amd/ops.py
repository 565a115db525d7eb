"""Tensor-level wrappers over the C ABI (one function per entry point of include/resdepth_hip.h).

Activations are NHWC fp32 tensors shaped [N, H, W, C]; weights keep the torch layouts of the
reference's modules (nn.Conv2d [Cout,Cin,3,3], nn.ConvTranspose2d [Cin,Cout,2,2],
lib/UNet.py:4-5,21) and are re-packed into GEMM operand layouts by `pack_*`.
Every call enqueues on torch's current HIP stream and never synchronises.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib
from ._lib import amax_slot, check, load, ptr, quant_next, slot_of, stream_ptr, tag, workspace

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def _f32(t, name):
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    return t


def _gemm_slots(x, w, out=None, out2=None, img_ok=False):
    """rd_quant_next for a GEMM launch: the magnitude slots of its two operands (both or neither: an operand without a slot
    sends the launch to the six-product body) and of its outputs.  Per-image slot arrays (inference, _lib.AmaxPool(per_image=))
    only go to the entry points that index them (img_ok), and never mixed with per-tensor slots of an activation."""
    a, b = slot_of(x), slot_of(w)
    if a is None or b is None or _lib.products() != 3:
        a = b = None
    kinds = {bool(getattr(t, "_rd_img", 0)) for t in (a, out, out2) if t is not None}
    if True in kinds:
        if not img_ok:
            return                                              # (amax_slot hands no per-image slot to such a caller: a is the only one)
        if False in kinds:
            a = b = None                                        # a per-tensor activation slot beside per-image outputs: six products
    quant_next(a, b, out, out2)


def _out_slot(img_ok=False):
    """slot for a tensor this launch produces, when a pool is active (the engine's forward / backward in split2h mode)"""
    return amax_slot(img_ok) if _lib.products() == 3 else None


def _weight_slot(device):
    """a zeroed slot of its own for a packed weight (the per-layer pack calls; the fused pack has one array for the model)"""
    return torch.zeros(_lib.AMAX_WORDS, dtype=torch.int32, device=device) if _lib.products() == 3 else None


def amax_of(x):
    """Tag `x` with a freshly computed magnitude slot (rd_amax: one extra pass over x) -- for operands no kernel of this library
    produced (tests, inputs of the generic first-convolution path).  No-op outside split2h mode."""
    if _lib.products() != 3 or slot_of(x) is not None:
        return x
    slot = torch.zeros(_lib.AMAX_WORDS, dtype=torch.int32, device=x.device)
    check(load().rd_amax(ptr(x), x.numel(), slot.data_ptr(), stream_ptr()), "amax")
    return tag(x, slot)


def _packed_buffer(rows, taps, cin, device):
    """Opaque packed GEMM operand (fp32 layout + split-bf16 layout, include/resdepth_hip.h); shape[0] == rows."""
    nbytes = load().rd_packed_weight_bytes(rows, taps, cin)
    cols = (nbytes + 4 * rows - 1) // (4 * rows)
    return torch.empty(rows, cols, device=device, dtype=torch.float32)


def pack_conv3x3_weight(w, need_dgrad=True):
    cout, cin = w.shape[0], w.shape[1]
    wf = _packed_buffer(cout, 9, cin, w.device)
    wd = _packed_buffer(cin, 9, cout, w.device) if need_dgrad else None
    slot = _weight_slot(w.device)
    quant_next(out2=slot)
    check(load().rd_pack_conv3x3_weight(ptr(w.detach()), ptr(wf), ptr(wd), cout, cin, stream_ptr()), "pack_conv3x3")
    return tag(wf, slot), tag(wd, slot)


def pack_conv3x3_weight_folded(w, row_scale):
    """Forward operand with eval-mode BatchNorm folded in: rows scaled by gamma / sqrt(running_var + eps)."""
    cout, cin = w.shape[0], w.shape[1]
    wf = _packed_buffer(cout, 9, cin, w.device)
    slot = _weight_slot(w.device)
    quant_next(out2=slot)
    check(load().rd_pack_conv3x3_weight_folded(ptr(w.detach()), ptr(_f32(row_scale, "row_scale")), ptr(wf), cout, cin,
                                               stream_ptr()), "pack_conv3x3_folded")
    return tag(wf, slot)


def pack_convt2x2_weight(w, need_dgrad=True):
    cin, cout = w.shape[0], w.shape[1]
    wtf = _packed_buffer(4 * cout, 1, cin, w.device)
    wtd = _packed_buffer(cin, 4, cout, w.device) if need_dgrad else None
    slot = _weight_slot(w.device)
    quant_next(out2=slot)
    check(load().rd_pack_convt2x2_weight(ptr(w.detach()), ptr(wtf), ptr(wtd), cin, cout, stream_ptr()), "pack_convt")
    return tag(wtf, slot), tag(wtd, slot)


def conv3x3_fwd(x, wf):
    n, h, w, cin = x.shape
    cout = wf.shape[0]
    z = torch.empty(n, h, w, cout, device=x.device, dtype=torch.float32)
    _gemm_slots(x, wf)
    check(load().rd_conv3x3_fwd(ptr(_f32(x, "x")), ptr(wf), ptr(z), n, h, w, cin, cout, stream_ptr()), "conv3x3_fwd")
    return z


def conv3x3_fwd_stats(x, wf):
    """-> (z, sums[2*Cout] float64): the BN batch statistics come out of the GEMM epilogue."""
    n, h, w, cin = x.shape
    cout = wf.shape[0]
    z = torch.empty(n, h, w, cout, device=x.device, dtype=torch.float32)
    sums = torch.empty(2 * cout, device=x.device, dtype=torch.float64)
    ws = workspace(load().rd_conv3x3_fwd_stats_ws_bytes(n, h, w, cin, cout), x.device)
    _gemm_slots(x, wf)
    check(load().rd_conv3x3_fwd_stats(ptr(_f32(x, "x")), ptr(wf), ptr(z), ptr(sums), n, h, w, cin, cout, ws.data_ptr(),
                                      ws.numel(), stream_ptr()), "conv3x3_fwd_stats")
    return z, sums


def conv3x3_fwd_act(x, wf_folded, shift, slope, pool=False):
    """Inference: -> (a, pooled | None) with a = act(conv(x) + shift); the 2x2 max-pool comes out of the same epilogue."""
    n, h, w, cin = x.shape
    cout = wf_folded.shape[0]
    a = torch.empty(n, h, w, cout, device=x.device, dtype=torch.float32)
    pooled = torch.empty(n, h // 2, w // 2, cout, device=x.device, dtype=torch.float32) if pool else None
    sa, sp = _out_slot(True), (_out_slot(True) if pool else None)
    _gemm_slots(x, wf_folded, sa, sp, img_ok=True)
    check(load().rd_conv3x3_fwd_act(ptr(_f32(x, "x")), ptr(wf_folded), ptr(shift), float(slope), ptr(a), ptr(pooled), n, h, w,
                                    cin, cout, stream_ptr()), "conv3x3_fwd_act")
    return tag(a, sa), tag(pooled, sp)


def conv3x3_fwd_bn(x, wf, running_mean, running_var, num_batches_tracked, eps=BN_EPS, momentum=BN_MOMENTUM):
    """-> (z, mean, invstd): convolution + training-mode BatchNorm statistics (incl. the running-statistics update) in two
    launches; the statistics come out of the GEMM epilogue."""
    n, h, w, cin = x.shape
    cout = wf.shape[0]
    z = torch.empty(n, h, w, cout, device=x.device, dtype=torch.float32)
    mean = torch.empty(cout, device=x.device, dtype=torch.float32)
    invstd = torch.empty(cout, device=x.device, dtype=torch.float32)
    ws = workspace(load().rd_conv3x3_fwd_stats_ws_bytes(n, h, w, cin, cout), x.device)
    _gemm_slots(x, wf)
    check(load().rd_conv3x3_fwd_bn(ptr(_f32(x, "x")), ptr(wf), ptr(z), float(n * h * w), eps, momentum, ptr(mean), ptr(invstd),
                                   ptr(running_mean), ptr(running_var), ptr(num_batches_tracked), n, h, w, cin, cout,
                                   ws.data_ptr(), ws.numel(), stream_ptr()), "conv3x3_fwd_bn")
    return z, mean, invstd


class BnHook:
    """The conv block whose activation gradient a data-gradient kernel produces: that kernel's epilogue then emits the
    block's BN-backward statistics (include/resdepth_hip.h, rd_*_bwd_data_bnstats).  mode 1: z at the gradient's
    resolution; mode 2: the gradient is the POOLED one and z holds the values at the arg-max positions (zpool)."""

    def __init__(self, z, mean, invstd, gamma, beta, slope, slope_dev=None, mode=1):
        self.z, self.mean, self.invstd, self.gamma, self.beta = z, mean, invstd, gamma.detach(), beta.detach()
        self.slope, self.slope_dev, self.mode = float(slope), slope_dev, mode

    def args(self):
        return (ptr(self.z), ptr(self.mean), ptr(self.invstd), ptr(self.gamma), ptr(self.beta), self.slope, ptr(self.slope_dev))


def _bn_part(pixels, c, device):
    return torch.empty(load().rd_bn_bwd_part_floats(pixels, c), device=device, dtype=torch.float32)


def conv3x3_bwd_data(dz, wd, bn=None):
    """bn (BnHook, optional): also return (partial rows tensor, row count) of the BN-backward statistics; row count 0 =
    this shape has no statistics epilogue."""
    n, h, w, cout = dz.shape
    cin = wd.shape[0]
    dx = torch.empty(n, h, w, cin, device=dz.device, dtype=torch.float32)
    so = _out_slot()
    tag(dx, so)
    _gemm_slots(dz, wd, so)
    if bn is None:
        check(load().rd_conv3x3_bwd_data(ptr(dz), ptr(wd), ptr(dx), n, h, w, cin, cout, stream_ptr()), "conv3x3_bwd_data")
        return dx
    part, rows = _bn_part(n * h * w, cin, dz.device), ctypes.c_int(0)
    check(load().rd_conv3x3_bwd_data_bnstats(ptr(dz), ptr(wd), ptr(dx), n, h, w, cin, cout, *bn.args(), bn.mode, ptr(part),
                                             part.numel(), ctypes.byref(rows), stream_ptr()), "conv3x3_bwd_data_bnstats")
    return dx, (part, rows.value)


def bn_bwd_stats_finalize(parts, c, dgamma=None, dbeta=None, dextra=None):
    """parts: one or two (partial rows, row count) pairs -> sums [4*C] float64 (layout of bn_act_bwd_reduce)."""
    (pa, ra), (pb, rb) = parts[0], (parts[1] if len(parts) > 1 else (None, 0))
    sums = torch.empty(4 * c, device=pa.device, dtype=torch.float64)
    check(load().rd_bn_bwd_stats_finalize(ptr(pa), ra, ptr(pb), rb, c, ptr(sums), ptr(dgamma), ptr(dbeta), ptr(dextra),
                                          stream_ptr()), "bn_bwd_stats_finalize")
    return sums


def conv3x3_bwd_weight(x, dz, out=None, ws_slot=0):
    n, h, w, cin = x.shape
    cout = dz.shape[3]
    if out is None:
        out = torch.empty(cout, cin, 3, 3, device=x.device, dtype=torch.float32)
    nb = load().rd_conv3x3_bwd_weight_ws_bytes(n, h, w, cin, cout)
    ws = workspace(nb, x.device, ws_slot)
    _gemm_slots(dz, x)
    check(load().rd_conv3x3_bwd_weight(ptr(x), ptr(dz), ptr(out), n, h, w, cin, cout, ws.data_ptr(), ws.numel(),
                                       stream_ptr()), "conv3x3_bwd_weight")
    return out


def conv3x3_first_fwd(x_nchw, w):
    n, cin, h, wd_ = x_nchw.shape
    cout = w.shape[0]
    z = torch.empty(n, h, wd_, cout, device=x_nchw.device, dtype=torch.float32)
    check(load().rd_conv3x3_first_fwd(ptr(_f32(x_nchw, "x")), ptr(w.detach()), ptr(z), n, h, wd_, cin, cout,
                                      stream_ptr()), "conv3x3_first_fwd")
    return z


def conv3x3_first_fwd_stats(x_nchw, w):
    n, cin, h, wd_ = x_nchw.shape
    cout = w.shape[0]
    z = torch.empty(n, h, wd_, cout, device=x_nchw.device, dtype=torch.float32)
    sums = torch.empty(2 * cout, device=x_nchw.device, dtype=torch.float64)
    ws = workspace(load().rd_conv3x3_first_fwd_stats_ws_bytes(n, h, wd_, cin, cout), x_nchw.device)
    check(load().rd_conv3x3_first_fwd_stats(ptr(_f32(x_nchw, "x")), ptr(w.detach()), ptr(z), ptr(sums), n, h, wd_, cin,
                                            cout, ws.data_ptr(), ws.numel(), stream_ptr()), "conv3x3_first_fwd_stats")
    return z, sums


def conv3x3_first_fwd_bn(x_nchw, w, running_mean, running_var, num_batches_tracked, eps=BN_EPS, momentum=BN_MOMENTUM):
    n, cin, h, wd_ = x_nchw.shape
    cout = w.shape[0]
    z = torch.empty(n, h, wd_, cout, device=x_nchw.device, dtype=torch.float32)
    mean = torch.empty(cout, device=z.device, dtype=torch.float32)
    invstd = torch.empty(cout, device=z.device, dtype=torch.float32)
    ws = workspace(load().rd_conv3x3_first_fwd_stats_ws_bytes(n, h, wd_, cin, cout), z.device)
    check(load().rd_conv3x3_first_fwd_bn(ptr(_f32(x_nchw, "x")), ptr(w.detach()), ptr(z), float(n * h * wd_), eps, momentum,
                                         ptr(mean), ptr(invstd), ptr(running_mean), ptr(running_var), ptr(num_batches_tracked),
                                         n, h, wd_, cin, cout, ws.data_ptr(), ws.numel(), stream_ptr()), "conv3x3_first_fwd_bn")
    return z, mean, invstd


def conv3x3_first_fwd_act_available(x_nchw, cout) -> bool:
    n, cin, h, w = x_nchw.shape
    return bool(load().rd_conv3x3_first_fwd_act_available(n, h, w, cin, cout))


def conv3x3_first_fwd_act(x_nchw, w, mean, invstd, gamma, beta, slope, slope_dev=None, pool=True):
    """Inference: first convolution + eval-mode BN + activation (+ MaxPool2d(2, 2)) in one kernel -> (a [N,H,W,Cout], pooled |
    None); the pre-BN tensor is never written (include/resdepth_hip.h: rd_conv3x3_first_fwd_act)."""
    n, cin, h, wd = x_nchw.shape
    cout = w.shape[0]
    a = torch.empty(n, h, wd, cout, device=x_nchw.device, dtype=torch.float32)
    pooled = torch.empty(n, h // 2, wd // 2, cout, device=x_nchw.device, dtype=torch.float32) if pool else None
    sp = _out_slot(True) if pool else None
    quant_next(out2=sp)
    tag(pooled, sp)
    check(load().rd_conv3x3_first_fwd_act(ptr(_f32(x_nchw, "x")), ptr(w.detach()), ptr(mean), ptr(invstd), ptr(gamma.detach()),
                                          ptr(beta.detach()), float(slope), ptr(slope_dev), ptr(a), ptr(pooled), n, h, wd, cin, cout,
                                          stream_ptr()), "conv3x3_first_fwd_act")
    return a, pooled


def conv3x3_first_bwd_weight_bn_available(x_nchw, cout) -> bool:
    n, cin, h, wd_ = x_nchw.shape
    return bool(load().rd_conv3x3_first_bwd_weight_bn_available(n, h, wd_, cin, int(cout)))


class LastConvGrad:
    """The full-resolution gradient operand g = conv3x3_last_bwd_data(dout, w_last) that is NOT materialised: consumers evaluate
    it from the network's 1-channel output gradient (conv3x3_first_bwd_weight_bn), or call .tensor() to build it after all."""

    def __init__(self, dout, w_last, c):
        self.dout, self.w_last, self.c = dout, w_last, c

    def tensor(self):
        return conv3x3_last_bwd_data(self.dout, self.w_last, self.c)


def conv3x3_first_bwd_weight_bn(x_nchw, z, mean, invstd, gamma, beta, slope, g_full, g_pool, idx, sums, count, training=True,
                                slope_dev=None, out=None, ws_slot=0):
    """Weight gradient of the first convolution with dz = bn_act_bwd_apply(z, ..., g_full, g_pool, idx, sums, count, training)
    evaluated on the fly (include/resdepth_hip.h: rd_conv3x3_first_bwd_weight_bn): no dz tensor.  g_full may be a LastConvGrad:
    then that operand is evaluated from dout too."""
    dout = w_last = None
    if isinstance(g_full, LastConvGrad):
        dout, w_last, g_full = g_full.dout, g_full.w_last.detach(), None
    n, cin, h, wd_ = x_nchw.shape
    cout = z.shape[3]
    if out is None:
        out = torch.empty(cout, cin, 3, 3, device=z.device, dtype=torch.float32)
    nb = load().rd_conv3x3_first_bwd_weight_ws_bytes(n, h, wd_, cin, cout)
    ws = workspace(nb, z.device, ws_slot)
    check(load().rd_conv3x3_first_bwd_weight_bn(ptr(x_nchw), ptr(z), ptr(mean), ptr(invstd), ptr(gamma.detach()), ptr(beta.detach()),
                                                float(slope), ptr(slope_dev), ptr(g_full), ptr(g_pool), ptr(idx), ptr(sums),
                                                float(count), 1 if training else 0, ptr(dout), ptr(w_last), ptr(out), n, h, wd_, cin,
                                                cout, ws.data_ptr(), ws.numel(), stream_ptr()), "conv3x3_first_bwd_weight_bn")
    return out


def conv3x3_first_bwd_weight(x_nchw, dz, out=None, ws_slot=0):
    n, cin, h, wd_ = x_nchw.shape
    cout = dz.shape[3]
    if out is None:
        out = torch.empty(cout, cin, 3, 3, device=dz.device, dtype=torch.float32)
    nb = load().rd_conv3x3_first_bwd_weight_ws_bytes(n, h, wd_, cin, cout)
    ws = workspace(nb, dz.device, ws_slot)
    check(load().rd_conv3x3_first_bwd_weight(ptr(x_nchw), ptr(dz), ptr(out), n, h, wd_, cin, cout, ws.data_ptr(),
                                             ws.numel(), stream_ptr()), "conv3x3_first_bwd_weight")
    return out


def conv3x3_last_fwd(s, w, bias, x_nchw):
    n, h, wd_, c = s.shape
    out = torch.empty(n, 1, h, wd_, device=s.device, dtype=torch.float32)
    xc = x_nchw.shape[1] if x_nchw is not None else 0
    check(load().rd_conv3x3_last_fwd(ptr(s), ptr(w.detach()), ptr(bias.detach() if bias is not None else None),
                                     ptr(x_nchw), xc, ptr(out), n, h, wd_, c, stream_ptr()), "conv3x3_last_fwd")
    return out


def conv3x3_last_bwd_data(dout, w, c, bn=None, write=True):
    """write=False (with bn): only the BN-backward statistics of the hook -> (None, (partial rows, count)); the gradient tensor
    itself is left to consumers that evaluate it from dout (LastConvGrad)."""
    n, _, h, wd_ = dout.shape
    ds = torch.empty(n, h, wd_, c, device=dout.device, dtype=torch.float32) if write else None
    if bn is None:
        check(load().rd_conv3x3_last_bwd_data(ptr(dout), ptr(w.detach()), ptr(ds), n, h, wd_, c, stream_ptr()),
              "conv3x3_last_bwd_data")
        return ds
    part, rows = _bn_part(n * h * wd_, c, dout.device), ctypes.c_int(0)
    check(load().rd_conv3x3_last_bwd_data_bnstats(ptr(dout), ptr(w.detach()), ptr(ds), n, h, wd_, c, *bn.args(), ptr(part),
                                                  part.numel(), ctypes.byref(rows), stream_ptr()),
          "conv3x3_last_bwd_data_bnstats")
    return ds, (part, rows.value)


# ---- the tail: last up-convolution composed with the last convolution (include/resdepth_hip.h) ------------------------------
def tail_available(cin, c0) -> bool:
    return bool(load().rd_tail_available(int(cin), int(c0)))


def tail_compose(wt_iohw, w_last, bias_t=None, forward=False):
    """-> (M [Cin, 4, 9], V [Cin, 16]): the last up-convolution's weight [Cin, C0, 2, 2] contracted with the last
    convolution's [1, C0, 3, 3] over C0.  forward=True: also VT [16, Cin, 1, 1] (V transposed, a 1x1 convolution weight) and
    B9 [9] (the up-convolution's bias seen through the last convolution)."""
    cin, c0 = wt_iohw.shape[0], wt_iohw.shape[1]
    m = torch.empty(cin, 4, 9, device=wt_iohw.device, dtype=torch.float32)
    v = torch.empty(cin, 16, device=wt_iohw.device, dtype=torch.float32)
    vt = torch.empty(16, cin, 1, 1, device=wt_iohw.device, dtype=torch.float32) if forward else None
    b9 = torch.empty(9, device=wt_iohw.device, dtype=torch.float32) if forward else None
    check(load().rd_tail_compose(ptr(wt_iohw.detach()), ptr(bias_t.detach() if bias_t is not None else None), ptr(w_last.detach()),
                                 ptr(m), ptr(v), ptr(vt), ptr(b9), cin, c0, stream_ptr()), "tail_compose")
    return (m, v, vt, b9) if forward else (m, v)


def conv3x3_last_fwd_tail(skip, t16, b9, w_last, bias, x_nchw):
    """Last convolution on s = up-convolution + act(BN(z)) without s (include/resdepth_hip.h: rd_conv3x3_last_fwd_tail).  skip:
    level 0's lazy-skip descriptor {z, mean, invstd, gamma, beta, slope, slope_dev}; t16 = conv1x1_fwd(x_coarse, VT)."""
    z = skip["z"]
    n, h, wd_, c = z.shape
    out = torch.empty(n, 1, h, wd_, device=z.device, dtype=torch.float32)
    xc = x_nchw.shape[1] if x_nchw is not None else 0
    beta = skip["beta"]
    check(load().rd_conv3x3_last_fwd_tail(ptr(z), ptr(skip["mean"]), ptr(skip["invstd"]), ptr(skip["gamma"].detach()),
                                          ptr(beta.detach()), float(skip["slope"]), ptr(skip["slope_dev"]), ptr(t16), ptr(b9),
                                          ptr(w_last.detach()), ptr(bias.detach() if bias is not None else None), ptr(x_nchw), xc,
                                          ptr(out), n, h, wd_, c, stream_ptr()), "conv3x3_last_fwd_tail")
    return out


def conv3x3_last_bwd_tail_fused(skip, dout, w_last):
    """One pass over level 0's z at the head of the backward: -> (wpartial, (BN-backward partial rows, count)) -- the partial
    sums of the last convolution's weight gradient (tail_wl_finish completes them) and the statistics of the level-0 hook."""
    z = skip["z"]
    n, h, wd_, c = z.shape
    nb = load().rd_conv3x3_last_bwd_tail_blocks(n, h, wd_)
    wpartial = torch.empty(nb, 9 * c + 9, device=z.device, dtype=torch.float64)
    part, rows = torch.empty(nb * 4 * c, device=z.device, dtype=torch.float32), ctypes.c_int(0)
    check(load().rd_conv3x3_last_bwd_tail_fused(ptr(z), ptr(skip["mean"]), ptr(skip["invstd"]), ptr(skip["gamma"].detach()),
                                                ptr(skip["beta"].detach()), float(skip["slope"]), ptr(skip["slope_dev"]), ptr(dout),
                                                ptr(w_last.detach()), ptr(wpartial), ptr(part), part.numel(), ctypes.byref(rows),
                                                n, h, wd_, c, stream_ptr()), "conv3x3_last_bwd_tail_fused")
    return wpartial, (part, rows.value)


def tail_wl_finish(wpartial, c16, wt_iohw, bias_t, dw=None, dbias=None, want_bias=True, ws_slot=0):
    """Completes conv3x3_last_bwd_tail_fused's weight-gradient partials with the up-convolution part (c16) and its bias."""
    c = (wpartial.shape[1] - 9) // 9
    if dw is None:
        dw = torch.empty(1, c, 3, 3, device=wpartial.device, dtype=torch.float32)
    if dbias is None and want_bias:
        dbias = torch.empty(1, device=wpartial.device, dtype=torch.float32)
    check(load().rd_tail_wl_finish(ptr(wpartial), wpartial.shape[0], ptr(c16), ptr(wt_iohw.detach()),
                                   ptr(bias_t.detach() if bias_t is not None else None), ptr(dw), ptr(dbias), wt_iohw.shape[0], c,
                                   stream_ptr()), "tail_wl_finish")
    return dw, dbias


def conv3x3_last_bwd_weight_tail(skip, dout, c16, wt_iohw, bias_t, dw=None, dbias=None, want_bias=True, ws_slot=0):
    """Weight / bias gradient of the last convolution whose input was never a tensor (conv3x3_last_fwd_tail): from z + dout,
    the correlations c16 of convt_last_bwd_weight and the up-convolution's weight / bias."""
    z = skip["z"]
    n, h, wd_, c = z.shape
    cin = wt_iohw.shape[0]
    if dw is None:
        dw = torch.empty(1, c, 3, 3, device=z.device, dtype=torch.float32)
    if dbias is None and want_bias:
        dbias = torch.empty(1, device=z.device, dtype=torch.float32)
    ws = workspace(load().rd_conv3x3_last_bwd_weight_tail_ws_bytes(n, h, wd_, c), z.device, ws_slot)
    check(load().rd_conv3x3_last_bwd_weight_tail(ptr(z), ptr(skip["mean"]), ptr(skip["invstd"]), ptr(skip["gamma"].detach()),
                                                 ptr(skip["beta"].detach()), float(skip["slope"]), ptr(skip["slope_dev"]), ptr(dout),
                                                 ptr(c16), ptr(wt_iohw.detach()), ptr(bias_t.detach() if bias_t is not None else None),
                                                 ptr(dw), ptr(dbias), n, h, wd_, cin, c, ws.data_ptr(), ws.numel(), stream_ptr()),
          "conv3x3_last_bwd_weight_tail")
    return dw, dbias


def convt_last_bwd_data(dout, v, bn=None):
    """Gradient w.r.t. the INPUT of the last up-convolution straight from the network's output gradient dout [N, 1, H, W]
    (16-tap stride-2 stencil V): == convt2x2_bwd_data(conv3x3_last_bwd_data(dout)) without the full-resolution tensor.
    bn: BnHook of the block that produced that input -> (dprev, (partial rows, count))."""
    n, _, h, w = dout.shape
    cin = v.shape[0]
    dprev = torch.empty(n, h // 2, w // 2, cin, device=dout.device, dtype=torch.float32)
    if bn is None:
        check(load().rd_convt_last_bwd_data(ptr(dout), ptr(v), ptr(dprev), n, h // 2, w // 2, cin, None, None, None, None, None, 0.0,
                                            None, None, 0, None, stream_ptr()), "convt_last_bwd_data")
        return dprev
    part, rows = _bn_part(n * (h // 2) * (w // 2), cin, dout.device), ctypes.c_int(0)
    check(load().rd_convt_last_bwd_data(ptr(dout), ptr(v), ptr(dprev), n, h // 2, w // 2, cin, *bn.args(), ptr(part), part.numel(),
                                        ctypes.byref(rows), stream_ptr()), "convt_last_bwd_data")
    return dprev, (part, rows.value)


def _desc_args(d):
    """(z, mean, invstd, gamma, beta, slope, slope_dev) pointers of a lazy activation descriptor (UNet._bn_forward)"""
    return (ptr(d["z"]), ptr(d["mean"]), ptr(d["invstd"]), ptr(d["gamma"].detach()), ptr(d["beta"].detach()), float(d["slope"]),
            ptr(d["slope_dev"]))


def tail_t16(x, v):
    """T [N, h, w, 16] = x . V for the forward tail (conv3x3_last_fwd_tail).  x: the up-convolution's input [N, h, w, Cin], or the
    lazy descriptor {z, mean, invstd, gamma, beta, slope, slope_dev} of the block that produces it (BN + activation on load)."""
    z = x["z"] if isinstance(x, dict) else x
    n, h, w, cin = z.shape
    t16 = torch.empty(n, h, w, 16, device=z.device, dtype=torch.float32)
    args = _desc_args(x) if isinstance(x, dict) else (ptr(_f32(z, "x")), None, None, None, None, 1.0, None)
    check(load().rd_tail_t16(*args, ptr(v), ptr(t16), n * h * w, cin, stream_ptr()), "tail_t16")
    return t16


def convt_last_bwd_weight(x, dout, w_last, out=None, c16=None, ws_slot=0):
    """Weight gradient [Cin, C0, 2, 2] of the last up-convolution from its input x [N, h, w, Cin] and the network's output
    gradient dout [N, 1, 2h, 2w]: == convt2x2_bwd_weight(x, conv3x3_last_bwd_data(dout)).  c16 (optional [Cin, 16] float64
    tensor): receives the correlations sum_p x[p][ci] dout[2p + d] (conv3x3_last_bwd_weight_tail takes them)."""
    z = x["z"] if isinstance(x, dict) else x            # dict: lazy descriptor of the producing block (see tail_t16)
    n, hc, wc, cin = z.shape
    c0 = w_last.shape[1]
    if out is None:
        out = torch.empty(cin, c0, 2, 2, device=z.device, dtype=torch.float32)
    ws = workspace(load().rd_convt_last_bwd_weight_ws_bytes(n, hc, wc, cin), z.device, ws_slot)
    args = _desc_args(x) if isinstance(x, dict) else (ptr(z), None, None, None, None, 1.0, None)
    check(load().rd_convt_last_bwd_weight_bn(*args, ptr(dout), ptr(w_last.detach()), ptr(out), ptr(c16), n, hc, wc, cin, c0,
                                             ws.data_ptr(), ws.numel(), stream_ptr()), "convt_last_bwd_weight")
    return out


def conv3x3_last_bwd_weight(s, dout, dw=None, dbias=None, want_bias=True, ws_slot=0):
    n, h, wd_, c = s.shape
    if dw is None:
        dw = torch.empty(1, c, 3, 3, device=s.device, dtype=torch.float32)
    if dbias is None and want_bias:
        dbias = torch.empty(1, device=s.device, dtype=torch.float32)
    nb = load().rd_conv3x3_last_bwd_weight_ws_bytes(n, h, wd_, c)
    ws = workspace(nb, s.device, ws_slot)
    check(load().rd_conv3x3_last_bwd_weight(ptr(s), ptr(dout), ptr(dw), ptr(dbias), n, h, wd_, c, ws.data_ptr(),
                                            ws.numel(), stream_ptr()), "conv3x3_last_bwd_weight")
    return dw, dbias


def convt2x2_fwd(x, wtf, bias, skip):
    n, h, w, cin = x.shape
    cout = wtf.shape[0] // 4
    out = torch.empty(n, 2 * h, 2 * w, cout, device=x.device, dtype=torch.float32)
    so = _out_slot(True)
    _gemm_slots(x, wtf, so, img_ok=True)
    check(load().rd_convt2x2_fwd(ptr(x), ptr(wtf), ptr(bias.detach() if bias is not None else None), ptr(skip),
                                 ptr(out), n, h, w, cin, cout, stream_ptr()), "convt2x2_fwd")
    return tag(out, so)


def convt2x2_fwd_bnskip(x, wtf, bias, z_skip, mean, invstd, gamma, beta, slope, slope_dev=None):
    """convT + bias + skip where skip = act(BN(z_skip)) is recomputed in the epilogue (the encoder level's activation is
    never materialised at full resolution)."""
    n, h, w, cin = x.shape
    cout = wtf.shape[0] // 4
    out = torch.empty(n, 2 * h, 2 * w, cout, device=x.device, dtype=torch.float32)
    so = _out_slot(True)
    tag(out, so)
    _gemm_slots(x, wtf, so, img_ok=True)
    check(load().rd_convt2x2_fwd_bnskip(ptr(x), ptr(wtf), ptr(bias.detach() if bias is not None else None), ptr(z_skip),
                                        ptr(mean), ptr(invstd), ptr(gamma.detach()), ptr(beta.detach()), float(slope),
                                        ptr(slope_dev), ptr(out), n, h, w, cin, cout, stream_ptr()), "convt2x2_fwd_bnskip")
    return out


def convt2x2_bwd_data(dout, wtd, bn=None):
    n, h2, w2, cout = dout.shape
    cin = wtd.shape[0]
    dx = torch.empty(n, h2 // 2, w2 // 2, cin, device=dout.device, dtype=torch.float32)
    _gemm_slots(dout, wtd)
    if bn is None:
        check(load().rd_convt2x2_bwd_data(ptr(dout), ptr(wtd), ptr(dx), n, h2 // 2, w2 // 2, cin, cout, stream_ptr()),
              "convt2x2_bwd_data")
        return dx
    part, rows = _bn_part(n * (h2 // 2) * (w2 // 2), cin, dout.device), ctypes.c_int(0)
    check(load().rd_convt2x2_bwd_data_bnstats(ptr(dout), ptr(wtd), ptr(dx), n, h2 // 2, w2 // 2, cin, cout, *bn.args(),
                                              ptr(part), part.numel(), ctypes.byref(rows), stream_ptr()),
          "convt2x2_bwd_data_bnstats")
    return dx, (part, rows.value)


def convt2x2_bwd_weight(x, dout, out=None, ws_slot=0):
    n, h, w, cin = x.shape
    cout = dout.shape[3]
    if out is None:
        out = torch.empty(cin, cout, 2, 2, device=x.device, dtype=torch.float32)
    nb = load().rd_convt2x2_bwd_weight_ws_bytes(n, h, w, cin, cout)
    ws = workspace(nb, x.device, ws_slot)
    _gemm_slots(dout, x)
    check(load().rd_convt2x2_bwd_weight(ptr(x), ptr(dout), ptr(out), n, h, w, cin, cout, ws.data_ptr(), ws.numel(),
                                        stream_ptr()), "convt2x2_bwd_weight")
    return out


# ---- bilinear up-mode (lib/UNet.py:17-24): conv1x1 on the coarse grid, then interpolate + bias + skip ---------
def pack_conv1x1_weight(w):
    """w: [Cout, Cin, 1, 1] (torch) -> packed (forward operand [Cout, Cin], data-gradient operand [Cin, Cout])."""
    cout, cin = w.shape[0], w.shape[1]
    wf = _packed_buffer(cout, 1, cin, w.device)
    wt = _packed_buffer(cin, 1, cout, w.device)
    slot = _weight_slot(w.device)
    quant_next(out2=slot)
    check(load().rd_pack_conv1x1_weight(ptr(w.detach()), ptr(wf), ptr(wt), cout, cin, stream_ptr()), "pack_conv1x1")
    return tag(wf, slot), tag(wt, slot)


def conv1x1_fwd(x, w2d):
    n, h, w, cin = x.shape
    cout = w2d.shape[0]
    out = torch.empty(n, h, w, cout, device=x.device, dtype=torch.float32)
    _gemm_slots(x, w2d)
    check(load().rd_conv1x1_fwd(ptr(_f32(x, "x")), ptr(w2d), ptr(out), n * h * w, cin, cout, stream_ptr()), "conv1x1_fwd")
    return out


def conv1x1_bwd_data(dy, wt):
    n, h, w, cout = dy.shape
    cin = wt.shape[0]
    dx = torch.empty(n, h, w, cin, device=dy.device, dtype=torch.float32)
    _gemm_slots(dy, wt)
    check(load().rd_conv1x1_bwd_data(ptr(dy), ptr(wt), ptr(dx), n * h * w, cin, cout, stream_ptr()), "conv1x1_bwd_data")
    return dx


def conv1x1_bwd_weight(x, dy, out=None, ws_slot=0):
    n, h, w, cin = x.shape
    cout = dy.shape[3]
    if out is None:
        out = torch.empty(cout, cin, 1, 1, device=x.device, dtype=torch.float32)
    nb = load().rd_conv1x1_bwd_weight_ws_bytes(n * h * w, cin, cout)
    ws = workspace(nb, x.device, ws_slot)
    _gemm_slots(dy, x)
    check(load().rd_conv1x1_bwd_weight(ptr(x), ptr(dy), ptr(out), n * h * w, cin, cout, ws.data_ptr(), ws.numel(),
                                       stream_ptr()), "conv1x1_bwd_weight")
    return out


def upsample2x_add_fwd(t, bias=None, skip=None):
    n, h, w, c = t.shape
    out = torch.empty(n, 2 * h, 2 * w, c, device=t.device, dtype=torch.float32)
    check(load().rd_upsample2x_add_fwd(ptr(t), ptr(bias.detach() if bias is not None else None), ptr(skip), ptr(out),
                                       n, h, w, c, stream_ptr()), "upsample2x_add_fwd")
    return out


def upsample2x_bwd(g):
    n, h2, w2, c = g.shape
    dt = torch.empty(n, h2 // 2, w2 // 2, c, device=g.device, dtype=torch.float32)
    check(load().rd_upsample2x_bwd(ptr(g), ptr(dt), n, h2 // 2, w2 // 2, c, stream_ptr()), "upsample2x_bwd")
    return dt


def channel_sum(g, out=None):
    c = g.shape[-1]
    pixels = g.numel() // c
    if out is None:
        out = torch.empty(c, device=g.device, dtype=torch.float32)
    nb = load().rd_channel_sum_ws_bytes(pixels, c)
    ws = workspace(nb, g.device)
    check(load().rd_channel_sum(ptr(g), ptr(out), pixels, c, ws.data_ptr(), ws.numel(), stream_ptr()), "channel_sum")
    return out


def bn_stats_partial(z):
    """-> sums [2*C] float64 (sum, sum of squares) over the pixels of z[N,H,W,C]."""
    c = z.shape[-1]
    pixels = z.numel() // c
    sums = torch.empty(2 * c, device=z.device, dtype=torch.float64)
    nb = load().rd_bn_stats_ws_bytes(pixels, c)
    ws = workspace(nb, z.device)
    check(load().rd_bn_stats_partial(ptr(z), ptr(sums), pixels, c, ws.data_ptr(), ws.numel(), stream_ptr()),
          "bn_stats_partial")
    return sums


def bn_stats_finalize(sums, count, running_mean, running_var, num_batches_tracked, eps=BN_EPS, momentum=BN_MOMENTUM):
    c = sums.numel() // 2
    mean = torch.empty(c, device=sums.device, dtype=torch.float32)
    invstd = torch.empty(c, device=sums.device, dtype=torch.float32)
    check(load().rd_bn_stats_finalize(ptr(sums), float(count), eps, momentum, ptr(mean), ptr(invstd),
                                      ptr(running_mean), ptr(running_var), ptr(num_batches_tracked), c, stream_ptr()),
          "bn_stats_finalize")
    return mean, invstd


def bn_eval_stats(running_mean, running_var, eps=BN_EPS):
    c = running_mean.numel()
    mean = torch.empty(c, device=running_mean.device, dtype=torch.float32)
    invstd = torch.empty(c, device=running_mean.device, dtype=torch.float32)
    check(load().rd_bn_eval_stats(ptr(running_mean), ptr(running_var), eps, ptr(mean), ptr(invstd), c, stream_ptr()),
          "bn_eval_stats")
    return mean, invstd


def bn_act_pool_fwd(z, mean, invstd, gamma, beta, slope, pool, slope_dev=None, want_a=True, want_zpool=False):
    """slope_dev: optional 1-element fp32 device tensor (nn.PReLU().weight) overriding `slope`.
    want_a=False (pooling only): the full-resolution activation is not written (returned as None).
    want_zpool (pooling only): also return z at the arg-max positions as a 4th value."""
    n, h, w, c = z.shape
    a = torch.empty_like(z) if (want_a or not pool) else None
    pooled = idx = zpool = None
    if pool:
        pooled = torch.empty(n, h // 2, w // 2, c, device=z.device, dtype=torch.float32)
        idx = torch.empty(n, h // 2, w // 2, c, device=z.device, dtype=torch.uint8)
        if want_zpool:
            zpool = torch.empty_like(pooled)
    sa, sp = (None, _out_slot()) if pool else (_out_slot(), None)       # the tensor the next GEMM takes as an operand
    quant_next(out=sa, out2=sp)
    tag(a, sa)
    tag(pooled, sp)
    check(load().rd_bn_act_pool_fwd(ptr(z), ptr(mean), ptr(invstd), ptr(gamma.detach()), ptr(beta.detach()),
                                    float(slope), ptr(slope_dev), ptr(a), ptr(pooled), ptr(idx), ptr(zpool), n, h, w, c,
                                    stream_ptr()),
          "bn_act_pool_fwd")
    if want_zpool:
        return a, pooled, idx, zpool
    return a, pooled, idx


def bn_act_bwd_reduce(z, mean, invstd, gamma, beta, slope, g_full, g_pool, idx, slope_dev=None, dgamma=None, dbeta=None,
                      dextra=None):
    """-> sums [4*C] float64: sum g', sum g'*xhat, sum g_full, sum_{y<=0} g*y (PReLU slope gradient).
    dgamma / dbeta / dextra (optional fp32 [C]): the affine-parameter gradients and sum g_full (the bias gradient of the
    transposed convolution feeding the skip add), written by the same reduction."""
    n, h, w, c = z.shape
    sums = torch.empty(4 * c, device=z.device, dtype=torch.float64)
    nb = load().rd_bn_act_bwd_ws_bytes(n, h, w, c)
    ws = workspace(nb, z.device)
    check(load().rd_bn_act_bwd_reduce(ptr(z), ptr(mean), ptr(invstd), ptr(gamma.detach()), ptr(beta.detach()),
                                      float(slope), ptr(slope_dev), ptr(g_full), ptr(g_pool), ptr(idx), ptr(sums),
                                      ptr(dgamma), ptr(dbeta), ptr(dextra), n, h, w, c, ws.data_ptr(), ws.numel(),
                                      stream_ptr()),
          "bn_act_bwd_reduce")
    return sums


def bn_act_bwd_apply(z, mean, invstd, gamma, beta, slope, g_full, g_pool, idx, sums, count, training=True,
                     dgamma=None, dbeta=None, slope_dev=None):
    n, h, w, c = z.shape
    dz = torch.empty_like(z)
    so = _out_slot()
    quant_next(out=so)
    tag(dz, so)
    check(load().rd_bn_act_bwd_apply(ptr(z), ptr(mean), ptr(invstd), ptr(gamma.detach()), ptr(beta.detach()),
                                     float(slope), ptr(slope_dev), ptr(g_full), ptr(g_pool), ptr(idx), ptr(sums),
                                     float(count),
                                     1 if training else 0, ptr(dz), ptr(dgamma), ptr(dbeta), n, h, w, c, stream_ptr()),
          "bn_act_bwd_apply")
    return dz


def masked_l1_partial(yp, y, mask, mean, std):
    """-> sums [2] float64 on device: (sum |p - t| over valid pixels, #valid)."""
    n = yp.shape[0]
    pps = yp.numel() // n
    sums = torch.empty(2, device=yp.device, dtype=torch.float64)
    nb = load().rd_masked_l1_ws_bytes(yp.numel())
    ws = workspace(nb, yp.device)
    check(load().rd_masked_l1_partial(ptr(yp), ptr(y), ptr(mask), ptr(mean), ptr(std), ptr(sums), n, pps,
                                      ws.data_ptr(), ws.numel(), stream_ptr()), "masked_l1_partial")
    return sums


def masked_l1_finish(yp, y, mask, mean, std, sums, numel_total, gout=None, want_loss=True, want_grad=True):
    """gout: optional 0-dim fp32 DEVICE tensor (upstream gradient); None means 1."""
    n = yp.shape[0]
    pps = yp.numel() // n
    loss = torch.empty(1, device=yp.device, dtype=torch.float32) if want_loss else None
    dyp = torch.empty_like(yp) if want_grad else None
    check(load().rd_masked_l1_finish(ptr(yp), ptr(y), ptr(mask), ptr(mean), ptr(std), ptr(sums), float(numel_total),
                                     ptr(gout), ptr(loss), ptr(dyp), n, pps, stream_ptr()), "masked_l1_finish")
    return loss, dyp


def adam_step(p, g, m, v, beta1, beta2, eps, weight_decay, step_size, bc2_sqrt, grad_scale=1.0):
    check(load().rd_adam_step(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), beta1, beta2, eps, weight_decay, step_size,
                              bc2_sqrt, grad_scale, stream_ptr()), "adam_step")


def adam_step_dev(p, g, m, v, scalars_dev):
    """rd_adam_step with its eight scalars in device memory (the launch a captured step replays; resdepth_amd/graph.py)."""
    assert scalars_dev.is_cuda and scalars_dev.dtype == torch.float32 and scalars_dev.numel() >= 8
    check(load().rd_adam_step_dev(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), ptr(scalars_dev), stream_ptr()), "adam_step_dev")


def sgd_step(p, g, buf, lr, weight_decay, momentum=0.0, dampening=0.0, nesterov=False, first_step=False, grad_scale=1.0):
    check(load().rd_sgd_step(ptr(p), ptr(g), ptr(buf), p.numel(), lr, weight_decay, momentum, dampening,
                             1 if nesterov else 0, 1 if first_step else 0, grad_scale, stream_ptr()), "sgd_step")


def nchw_to_nhwc(x):
    n, c, h, w = x.shape
    out = torch.empty(n, h, w, c, device=x.device, dtype=torch.float32)
    check(load().rd_nchw_to_nhwc(ptr(x), ptr(out), n, c, h, w, stream_ptr()), "nchw_to_nhwc")
    return out


def nhwc_to_nchw(x):
    n, h, w, c = x.shape
    out = torch.empty(n, c, h, w, device=x.device, dtype=torch.float32)
    check(load().rd_nhwc_to_nchw(ptr(x), ptr(out), n, c, h, w, stream_ptr()), "nhwc_to_nchw")
    return out


def blend_accumulate(pred, mean, std, pos, reg, tile_size, stride, raster):
    """raster (float64 [rows, cols], device) += blend-weighted de-normalised tiles; pos/reg int32 device tensors."""
    n = pred.shape[0]
    rows, cols = raster.shape
    if raster.dtype != torch.float64 or pos.dtype != torch.int32 or reg.dtype != torch.int32:
        raise TypeError("blend_accumulate: raster must be float64, pos/reg int32")
    check(load().rd_blend_accumulate(ptr(_f32(pred, "pred")), ptr(mean), ptr(std), ptr(pos), ptr(reg), n, tile_size,
                                     stride, ptr(raster), rows, cols, stream_ptr()), "blend_accumulate")
    return raster
