"""Diagnostic (GPU box): per-tensor gradient error of the HIP path and of the fp32 oracle, both against an
fp64 run of the oracle, at the full cfg-S architecture (N=2)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import unet_oracle as O
from resdepth_amd import UNet, masked_l1_loss

kw = dict(n_input_channels=3, start_kernel=64, depth=5, bias_conv_layer=True)
spec = O.Spec(**kw)
torch.manual_seed(0)
model = UNet(**kw)
sd0 = {k: v.clone() for k, v in model.state_dict().items()}
g = torch.Generator().manual_seed(1234)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
x = torch.randn(n, 3, 256, 256, generator=g)
y = x[:, 0:1] + 0.3 * torch.randn(n, 1, 256, 256, generator=g)
mask = torch.rand(n, 1, 256, 256, generator=g) > 0.05
mean = torch.randn(n, generator=g, dtype=torch.float64) * 50
std = torch.rand(n, generator=g) * 2.0 + 1.0

def run_oracle(dtype):
    leaves = {k: sd0[k].to(dtype).clone().requires_grad_(True) for k in O.param_keys(spec)}
    work = {k: (v.to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
    work.update(leaves)
    keep = {}
    yo = O.forward(work, x.to(dtype), spec, training=True, keep=keep)
    if dtype == torch.float64:
        p = yo * std.double().view(-1, 1, 1, 1) + mean.view(-1, 1, 1, 1)
        t = y.double() * std.double().view(-1, 1, 1, 1) + mean.view(-1, 1, 1, 1)
        lo = ((p - t).abs() * mask).sum() / mask.sum()
    else:
        lo = O.masked_l1_loss(yo, y, mask, mean, std)
    go = torch.autograd.grad(lo, list(leaves.values()))
    return yo.detach(), float(lo), dict(zip(leaves, go)), keep

y64, l64, g64, k64 = run_oracle(torch.float64)
y32, l32, g32, k32 = run_oracle(torch.float32)
model = model.to("cuda:0").train()
yp = model(x.cuda())
loss = masked_l1_loss(yp, y, mask, mean, std)
loss.backward()
rl = lambda a, b: float((a.double().cpu() - b.double()).norm() / (b.double().norm() + 1e-30))
print(f"loss hip {float(loss):.8f} o32 {l32:.8f} o64 {l64:.8f}")
print(f"fwd max|hip-o64| {float((yp.detach().cpu().double()-y64).abs().max()):.3e}  max|o32-o64| {float((y32.double()-y64).abs().max()):.3e}")
print(f"{'tensor':34s} {'hip vs o64':>12s} {'o32 vs o64':>12s} {'hip vs o32':>12s}")
for k, p in model.named_parameters():
    print(f"{k:34s} {rl(p.grad, g64[k]):12.3e} {rl(g32[k], g64[k]):12.3e} {rl(p.grad, g32[k]):12.3e}")

# ---- discrete decisions: ReLU-mask and pool-argmax flips between the HIP forward and the fp32 oracle
with torch.no_grad():
    out, S = model._engine_forward(x.cuda(), True, save=True)
for i, e in enumerate(S["enc"]):
    z = e["z"].permute(0, 3, 1, 2).cpu()
    zo = k32[f"z{i}"]
    bn = model.encoder[i][0][1]
    a_h = ((e["z"] - e["mean"]) * e["invstd"] * bn.weight + bn.bias).permute(0, 3, 1, 2).cpu()
    flips = int(((a_h > 0) != (k32[f"a{i}"] > 0)).sum())
    io = k32[f"idx{i}"]
    W = zo.shape[-1]
    pos = ((io // W) % 2) * 2 + (io % W) % 2
    pflips = int((e["idx"].permute(0, 3, 1, 2).cpu().long() != pos).sum())
    print(f"enc{i}: max|z-zo| {float((z-zo).abs().max()):.2e}  relu flips {flips} / {zo.numel()}  pool flips {pflips} / {io.numel()}")

# ---- identity activation (slope 1): no mask decisions -> errors should be pure fp32 rounding
import resdepth_amd.unet as U
U._SLOPES["relu"] = 1.0
O_slope = O._slope
O._slope = lambda name: 1.0
import torch.nn.functional as F
_lr = F.leaky_relu
y64, l64, g64, _ = run_oracle(torch.float64)
y32, l32, g32, _ = run_oracle(torch.float32)
torch.manual_seed(0)
model2 = UNet(**kw).to("cuda:0").train()
yp = model2(x.cuda())
loss = masked_l1_loss(yp, y, mask, mean, std)
loss.backward()
print("identity activation:")
print(f"fwd max|hip-o64| {float((yp.detach().cpu().double()-y64).abs().max()):.3e}  max|o32-o64| {float((y32.double()-y64).abs().max()):.3e}")
worst = 0
for k, p in model2.named_parameters():
    r = rl(p.grad, g64[k]); worst = max(worst, r)
    if "weight" in k and ("0.0.weight" in k or "1.0.weight" in k or "bottleneck.0" in k):
        print(f"{k:34s} {r:12.3e} {rl(g32[k], g64[k]):12.3e}")
print("worst hip vs o64:", worst)
