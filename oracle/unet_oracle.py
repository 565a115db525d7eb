"""Functional torch-CPU restatement of the ResDepth hot path (TEST INFRASTRUCTURE ONLY).

Every function cites the reference lines (relative to /root/reference) it restates.
The reference composes torch.nn modules; this file states the same arithmetic as a
flat sequence of torch.nn.functional calls over a state_dict with the reference's
key names, and exposes every intermediate tensor so per-kernel parity tests can
compare against them.  Pinned against fixtures generated from the reference itself
(tests/golden/make_golden.py -> tests/test_oracle_golden.py).

Supported variant = the reference's default architecture (lib/config.py:25-54):
relu|lrelu activations, up_mode='transpose', do_BN=True, outer_skip in {True, False},
outer_skip_BN in {True, False}, bias on the last conv optional.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

BN_EPS = 1e-5        # torch.nn.BatchNorm2d default, used by lib/UNet.py:45,66,86
BN_MOMENTUM = 0.1    # idem


@dataclass(frozen=True)
class Spec:
    """Constructor arguments of the reference UNet (lib/UNet.py:105-107)."""
    n_input_channels: int = 1
    start_kernel: int = 64
    max_filter_depth: int = 512
    depth: int = 8
    act_fn_encoder: str = "relu"
    act_fn_decoder: str = "relu"
    act_fn_bottleneck: str = "relu"
    do_BN: bool = True
    bias_conv_layer: bool = False
    outer_skip: bool = True
    outer_skip_BN: bool = False
    up_mode: str = "transpose"      # 'transpose' | 'bilinear' (lib/UNet.py:17-24)

    @property
    def filter_depths(self) -> List[int]:
        # lib/UNet.py:152-155 -- 64*2^i capped at max_filter_depth
        return [min(self.start_kernel * (2 ** i), self.max_filter_depth) for i in range(self.depth)]


def _slope(name: str):
    # lib/UNet.py:27-33 -- ReLU / LeakyReLU(default negative_slope 0.01) / PReLU (learnable, one parameter, init 0.25)
    if name == "relu":
        return 0.0
    if name == "lrelu":
        return 0.01
    if name == "prelu":
        return "prelu"
    raise ValueError(f"oracle supports relu|lrelu|prelu, got {name!r}")


def param_layout(spec: Spec):
    """Ordered (key, shape, kind) list of the reference state_dict (lib/UNet.py:157-194).

    kind in {'param', 'buffer'}; order == torch's state_dict() order of the reference
    module tree: encoder.{i}.0.0 conv / .0.1 BN, bottleneck.0/.1, decoder.{i}.0 convT,
    decoder.{i}.1.0 conv / .1.1 BN, decoder.{depth-1} convT, last_layer,
    layer_outer_skip.0 (iff outer_skip_BN).
    """
    fd = spec.filter_depths
    out = []

    def conv(prefix, cout, cin):
        out.append((prefix + ".weight", (cout, cin, 3, 3), "param"))
        if not spec.do_BN:                 # conv_block without BN carries the bias (lib/UNet.py:48-52)
            out.append((prefix + ".bias", (cout,), "param"))

    def bn(prefix, c):
        if not spec.do_BN and not prefix.startswith("layer_outer_skip"):
            return
        out.append((prefix + ".weight", (c,), "param"))
        out.append((prefix + ".bias", (c,), "param"))
        out.append((prefix + ".running_mean", (c,), "buffer"))
        out.append((prefix + ".running_var", (c,), "buffer"))
        out.append((prefix + ".num_batches_tracked", (), "buffer"))

    ai = 2 if spec.do_BN else 1          # index of the activation module inside a conv block

    def act(prefix, name):
        if name == "prelu":
            out.append((f"{prefix}.{ai}.weight", (1,), "param"))

    cin = spec.n_input_channels
    for i, c in enumerate(fd):
        conv(f"encoder.{i}.0.0", c, cin)
        bn(f"encoder.{i}.0.1", c)
        act(f"encoder.{i}.0", spec.act_fn_encoder)
        cin = c
    conv("bottleneck.0", fd[-1], fd[-1])
    bn("bottleneck.1", fd[-1])
    act("bottleneck", spec.act_fn_bottleneck)
    up = list(reversed(fd))

    def upconv(prefix, c):
        if spec.up_mode == "bilinear":     # Sequential(Upsample, conv1x1): the conv is module 1 (lib/UNet.py:20)
            out.append((prefix + ".1.weight", (c, c, 1, 1), "param"))
            out.append((prefix + ".1.bias", (c,), "param"))
        else:
            out.append((prefix + ".weight", (c, c, 2, 2), "param"))
            out.append((prefix + ".bias", (c,), "param"))

    for i, (ci, co) in enumerate(zip(up[:-1], up[1:])):
        upconv(f"decoder.{i}.0", ci)
        conv(f"decoder.{i}.1.0", co, ci)
        bn(f"decoder.{i}.1.1", co)
        act(f"decoder.{i}.1", spec.act_fn_decoder)
    upconv(f"decoder.{spec.depth - 1}", up[-1])
    out.append(("last_layer.weight", (1, spec.start_kernel, 3, 3), "param"))
    if spec.bias_conv_layer:
        out.append(("last_layer.bias", (1,), "param"))
    if spec.outer_skip and spec.outer_skip_BN:
        bn("layer_outer_skip.0", 1)
    return out


def init_state_dict(spec: Spec, seed: int) -> Dict[str, torch.Tensor]:
    """Default-initialised weights with the reference's RNG draw order.

    lib/UNet.py:17-24: every `upconv()` call builds a ModuleDict holding BOTH the
    bilinear branch (whose conv1x1 draws weight+bias from the global RNG and is then
    discarded) and the ConvTranspose2d.  Reproducing the draw order is what makes
    `torch.manual_seed(s); UNet(...)` give identical weights.
    """
    torch.manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    fd = spec.filter_depths

    def conv(prefix, cin, cout, bias):
        bias = bias or (not spec.do_BN and prefix != "last_layer")
        m = torch.nn.Conv2d(cin, cout, 3, 1, 1, bias=bias)
        sd[prefix + ".weight"] = m.weight.detach().clone()
        if bias:
            sd[prefix + ".bias"] = m.bias.detach().clone()

    def bn(prefix, c):
        if not spec.do_BN and not prefix.startswith("layer_outer_skip"):
            return
        sd[prefix + ".weight"] = torch.ones(c)
        sd[prefix + ".bias"] = torch.zeros(c)
        sd[prefix + ".running_mean"] = torch.zeros(c)
        sd[prefix + ".running_var"] = torch.ones(c)
        sd[prefix + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)

    def upconv(prefix, c):
        c1 = torch.nn.Conv2d(c, c, 1, 1)                  # conv1x1 of the bilinear branch: drawn first, always
        m = torch.nn.ConvTranspose2d(c, c, 2, 2)          # ... then the transposed conv; one of the two is discarded
        if spec.up_mode == "bilinear":
            prefix, m = prefix + ".1", c1
        sd[prefix + ".weight"] = m.weight.detach().clone()
        sd[prefix + ".bias"] = m.bias.detach().clone()

    cin = spec.n_input_channels
    for i, c in enumerate(fd):
        conv(f"encoder.{i}.0.0", cin, c, False)
        bn(f"encoder.{i}.0.1", c)
        cin = c
    conv("bottleneck.0", fd[-1], fd[-1], False)
    bn("bottleneck.1", fd[-1])
    up = list(reversed(fd))
    for i, (ci, co) in enumerate(zip(up[:-1], up[1:])):
        upconv(f"decoder.{i}.0", ci)
        conv(f"decoder.{i}.1.0", ci, co, False)
        bn(f"decoder.{i}.1.1", co)
    upconv(f"decoder.{spec.depth - 1}", up[-1])
    conv("last_layer", spec.start_kernel, 1, spec.bias_conv_layer)
    if spec.outer_skip and spec.outer_skip_BN:
        bn("layer_outer_skip.0", 1)
    for k, shape, _ in param_layout(spec):           # PReLU slopes: constant init 0.25, no RNG draw
        if k not in sd:
            sd[k] = torch.full(shape, 0.25)
    # return in state_dict order
    return {k: sd[k] for k, _, _ in param_layout(spec)}


def _bn_act(z, sd, prefix, slope, training, update_running, mask=None):
    """conv output -> BatchNorm2d -> (Leaky)ReLU   (lib/UNet.py:44-47, 65-68, 85-87).

    `mask` (bool, optional) imposes the activation's branch decision (True = positive branch) instead of
    deriving it from the sign of the BN output: used by the parity tests to compare gradients under
    IDENTICAL discrete decisions (one flipped ReLU in a 10^6-element layer moves rel-L2 by 1e-3)."""
    if prefix + ".running_mean" in sd:
        rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
        if training and not update_running:
            rm, rv = rm.clone(), rv.clone()
        y = F.batch_norm(z, rm, rv, sd[prefix + ".weight"], sd[prefix + ".bias"],
                         training=training, momentum=BN_MOMENTUM, eps=BN_EPS)
        if training and update_running:
            sd[prefix + ".num_batches_tracked"] += 1
    else:                                  # do_BN=False: the conv's own bias was already added by the caller
        y = z
    if slope == "prelu":
        w = sd[prefix.rsplit(".", 1)[0] + (".2" if prefix + ".running_mean" in sd else ".1") + ".weight"]
        return torch.where(mask, y, y * w) if mask is not None else F.prelu(y, w)
    if mask is not None:
        return torch.where(mask, y, y * slope)
    return F.leaky_relu(y, slope) if slope != 0.0 else F.relu(y)


def forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, spec: Spec, training: bool = True,
            update_running: bool = True, keep: Optional[dict] = None,
            decisions: Optional[dict] = None) -> torch.Tensor:
    """UNet.forward (lib/UNet.py:196-246) as a flat functional graph.

    `decisions` (optional) imposes discrete choices: mask_e{i} / mask_b / mask_d{i} (bool activations masks)
    and idx{i} (pool arg-max, flat indices) -- see _bn_act.
    `keep`, if given, receives every intermediate (NCHW): z{i} conv outputs, a{i}
    post-activation skips, p{i}/idx{i} pooled values and flat argmax indices, zb/ab
    bottleneck, u{i}/s{i} up-conv output / skip sum, zd{i}/ad{i} decoder conv blocks,
    'res' last conv output.
    """
    k = keep if keep is not None else {}
    dec = decisions if decisions is not None else {}
    d = spec.depth
    se, sb, sdec = _slope(spec.act_fn_encoder), _slope(spec.act_fn_bottleneck), _slope(spec.act_fn_decoder)
    skips = []
    out = x
    for i in range(d):                                            # lib/UNet.py:201-207
        z = F.conv2d(out, sd[f"encoder.{i}.0.0.weight"], sd.get(f"encoder.{i}.0.0.bias"), 1, 1)
        a = _bn_act(z, sd, f"encoder.{i}.0.1", se, training, update_running, dec.get(f"mask_e{i}"))
        skips.append(a)
        if f"idx{i}" in dec:          # imposed arg-max (flat H*W indices, as torch returns them)
            idx = dec[f"idx{i}"]
            out = a.flatten(2).gather(2, idx.flatten(2)).view(idx.shape)
        else:
            out, idx = F.max_pool2d(a, 2, 2, return_indices=True)
        k[f"z{i}"], k[f"a{i}"], k[f"p{i}"], k[f"idx{i}"] = z, a, out, idx
    z = F.conv2d(out, sd["bottleneck.0.weight"], sd.get("bottleneck.0.bias"), 1, 1)       # lib/UNet.py:210
    out = _bn_act(z, sd, "bottleneck.1", sb, training, update_running, dec.get("mask_b"))
    k["zb"], k["ab"] = z, out
    for i in range(d):                                            # lib/UNet.py:213-224
        pre = f"decoder.{i}.0" if i < d - 1 else f"decoder.{i}"
        if spec.up_mode == "bilinear":    # nn.Upsample(mode='bilinear', scale_factor=2) -> conv1x1  (lib/UNet.py:20)
            u = F.conv2d(F.interpolate(out, scale_factor=2, mode="bilinear"), sd[pre + ".1.weight"], sd[pre + ".1.bias"])
        else:
            u = F.conv_transpose2d(out, sd[pre + ".weight"], sd[pre + ".bias"], stride=2)
        s = skips[-1 - i] + u                                     # SkipConnection: ADD, lib/UNet.py:100-101
        k[f"u{i}"], k[f"s{i}"] = u, s
        if i < d - 1:
            z = F.conv2d(s, sd[f"decoder.{i}.1.0.weight"], sd.get(f"decoder.{i}.1.0.bias"), 1, 1)
            out = _bn_act(z, sd, f"decoder.{i}.1.1", sdec, training, update_running, dec.get(f"mask_d{i}"))
            k[f"zd{i}"], k[f"ad{i}"] = z, out
        else:
            out = s
    res = F.conv2d(out, sd["last_layer.weight"], sd.get("last_layer.bias"), 1, 1)   # lib/UNet.py:227
    k["res"] = res
    if spec.outer_skip:                                           # lib/UNet.py:230-244
        x0 = x[:, 0:1]
        if spec.outer_skip_BN:
            rm, rv = sd["layer_outer_skip.0.running_mean"], sd["layer_outer_skip.0.running_var"]
            if training and not update_running:
                rm, rv = rm.clone(), rv.clone()
            x0 = F.batch_norm(x0, rm, rv, sd["layer_outer_skip.0.weight"], sd["layer_outer_skip.0.bias"],
                              training=training, momentum=BN_MOMENTUM, eps=BN_EPS)
            if training and update_running:
                sd["layer_outer_skip.0.num_batches_tracked"] += 1
        res = x0 + res
    return res


def masked_l1_loss(y_pred, y, loss_mask, mean, std, sign=None):
    """Trainer._compute_denormalized_loss (lib/Trainer.py:87-100) with
    denormalize_torch (lib/data_normalization.py:29-38) folded in.

    `sign` (optional, same shape as y_pred, values in {-1, 0, +1}) imposes the discrete decision of the L1 loss --
    sign(p - t) per pixel -- instead of deriving it from this function's own p - t: |d| is evaluated as sign * d.
    Used like `decisions=` in forward(): one pixel whose residual is at rounding level flips the sign of its gradient
    and, at batch 32 (2 M pixels), moves the weakly coherent deep-layer gradients by ~1e-3 rel-L2.

    p_i = y_pred_i * std_i + mean_i (two roundings, per sample), same for y; both are
    zeroed where loss_mask == 0; L1Loss(mean) over ALL elements; then
    loss * numel / sum(mask).
    """
    mean = torch.as_tensor(mean).flatten()
    std = torch.as_tensor(std).flatten()
    m32 = torch.tensor(mean.tolist(), dtype=torch.float32).view(-1, 1, 1, 1)
    s32 = torch.tensor(std.tolist(), dtype=torch.float32).view(-1, 1, 1, 1)
    p = y_pred * s32 + m32
    t = y * s32 + m32
    valid = loss_mask != 0
    p = torch.where(valid, p, torch.zeros_like(p))
    t = torch.where(valid, t, torch.zeros_like(t))
    loss = ((p - t) * sign.to(p.dtype)).mean() if sign is not None else (p - t).abs().mean()
    return loss * loss_mask.numel() / loss_mask.sum()


def adam_step(params: List[torch.Tensor], grads: List[torch.Tensor], exp_avg: List[torch.Tensor],
              exp_avg_sq: List[torch.Tensor], step: int, lr: float = 2e-4, betas=(0.9, 0.999),
              eps: float = 1e-8, weight_decay: float = 1e-5) -> None:
    """One torch.optim.Adam step (lib/utils.py:329-331; defaults lib/config.py:100-103).

    Classic coupled-L2 Adam, in place; `step` is the 1-based step number AFTER the
    increment.  Restated from the published algorithm (torch.optim.Adam, amsgrad=False,
    maximize=False); pinned against torch.optim.Adam in tests/test_oracle_golden.py.
    """
    b1, b2 = betas
    bc1 = 1.0 - b1 ** step
    bc2 = 1.0 - b2 ** step
    step_size = lr / bc1
    bc2_sqrt = math.sqrt(bc2)
    for p, g, m, v in zip(params, grads, exp_avg, exp_avg_sq):
        if weight_decay != 0.0:
            g = g + weight_decay * p
        m.lerp_(g, 1.0 - b1)
        v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
        denom = (v.sqrt() / bc2_sqrt).add_(eps)
        p.addcdiv_(m, denom, value=-step_size)


def sgd_step(params: List[torch.Tensor], grads: List[torch.Tensor], bufs: List[Optional[torch.Tensor]], lr: float,
             weight_decay: float = 0.0, momentum: float = 0.0, dampening: float = 0.0, nesterov: bool = False) -> None:
    """One torch.optim.SGD step (lib/utils.py:332-334 builds SGD(lr, weight_decay); momentum / dampening / nesterov are
    torch's further options), in place.  Restated from the published algorithm (torch.optim.SGD, maximize=False):
    g += wd * p;  momentum: buf = g on the first step, else buf = momentum * buf + (1 - dampening) * g, and
    g = g + momentum * buf (nesterov) or buf;  p -= lr * g.  `bufs[i] is None` marks "no momentum buffer yet"; the list is
    updated in place.  Pinned against torch.optim.SGD through tests/golden/g16_sgd.npz / g17_sgd_mom.npz."""
    for i, (p, g) in enumerate(zip(params, grads)):
        if weight_decay != 0.0:
            g = g + weight_decay * p
        if momentum != 0.0:
            if bufs[i] is None:
                bufs[i] = g.clone()
            else:
                bufs[i].mul_(momentum).add_(g, alpha=1.0 - dampening)
            g = g + momentum * bufs[i] if nesterov else bufs[i]
        p.add_(g, alpha=-lr)


def param_keys(spec: Spec) -> List[str]:
    return [k for k, _, kind in param_layout(spec) if kind == "param"]


def train_step(sd, batch, spec: Spec, opt_state: dict, lr=2e-4, betas=(0.9, 0.999), eps=1e-8,
               weight_decay=1e-5, keep: Optional[dict] = None, sgd: Optional[dict] = None):
    """One reference training iteration: Trainer.inference_one_batch('train') +
    optimizer.step() (lib/Trainer.py:159-199, 212-222).  Returns (loss, grads dict).

    `opt_state` = {'step': int, 'exp_avg': {key: t}, 'exp_avg_sq': {key: t}} (created
    lazily like torch.optim.Adam does).  `sgd` = {'momentum', 'nesterov'} switches to torch.optim.SGD(lr,
    weight_decay, ...) (opt_state then holds 'buf').  With `keep`, keep['grad_input'] = d loss / d input.
    """
    keys = param_keys(spec)
    leaves = {k_: sd[k_].detach().clone().requires_grad_(True) for k_ in keys}
    work = dict(sd)
    work.update(leaves)
    x_in = batch["input"].detach().clone().requires_grad_(True) if keep is not None else batch["input"]
    y_pred = forward(work, x_in, spec, training=True, update_running=True, keep=keep)
    # BN buffers are updated in place through `work` (it shares sd's buffer tensors)
    loss = masked_l1_loss(y_pred, batch["target"], batch["loss_mask"], batch["dsm_mean"], batch["dsm_std"])
    glist = torch.autograd.grad(loss, [leaves[k_] for k_ in keys] + ([x_in] if keep is not None else []))
    grads = dict(zip(keys, glist))
    if keep is not None:
        keep["y_pred"] = y_pred.detach()
        keep["grad_input"] = glist[-1]
    if sgd is not None:
        if not opt_state:
            opt_state.update(step=0, buf=[None] * len(keys))
        opt_state["step"] += 1
        with torch.no_grad():
            sgd_step([sd[k_] for k_ in keys], [grads[k_] for k_ in keys], opt_state["buf"], lr, weight_decay,
                     sgd.get("momentum", 0.0), sgd.get("dampening", 0.0), sgd.get("nesterov", False))
        return float(loss.detach()), grads
    if not opt_state:
        opt_state.update(step=0, exp_avg={k_: torch.zeros_like(sd[k_]) for k_ in keys},
                         exp_avg_sq={k_: torch.zeros_like(sd[k_]) for k_ in keys})
    opt_state["step"] += 1
    with torch.no_grad():
        adam_step([sd[k_] for k_ in keys], [grads[k_] for k_ in keys],
                  [opt_state["exp_avg"][k_] for k_ in keys], [opt_state["exp_avg_sq"][k_] for k_ in keys],
                  opt_state["step"], lr, betas, eps, weight_decay)
    return float(loss.detach()), grads


def synthetic_batch(n: int, c: int, t: int, seed: int = 1234, nodata_frac: float = 0.05):
    """Synthetic DataLoader-shaped batch: keys as produced by
    DsmOrthoDataset.__getitem__ + default collate (lib/DsmOrthoDataset.py:281-291)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, c, t, t, generator=g)
    y = x[:, 0:1] + 0.3 * torch.randn(n, 1, t, t, generator=g)
    mask = torch.rand(n, 1, t, t, generator=g) > nodata_frac
    return {
        "input": x, "target": y, "loss_mask": mask,
        "dsm_mean": torch.randn(n, generator=g, dtype=torch.float64) * 50.0,
        "dsm_std": torch.full((n,), 3.0),
    }
