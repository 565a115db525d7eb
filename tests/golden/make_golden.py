#!/usr/bin/env python3
"""Generate golden fixtures from the REFERENCE implementation (build container only).

Run:  python tests/golden/make_golden.py      (needs /root/reference; never runs on the GPU box)

Imports the reference's own `lib/UNet.py` (the only hot-path module that imports on
this toolchain) and `lib/data_normalization.denormalize_torch` (behind
an empty `torchvision` stand-in module object, because that file imports torchvision
at the top for an unrelated function).  `lib/Trainer.py` cannot be imported
(tensorboard/easydict/osgeo are absent), so the 6 lines of
Trainer._compute_denormalized_loss (lib/Trainer.py:87-100) are driven here through
the reference's denormalize_torch + torch.nn.L1Loss(reduction='mean')
(lib/utils.py:285) + torch.optim.Adam(lr, weight_decay) (lib/utils.py:329-331).

Outputs (data only -- inputs and expected outputs, no reference source):
  g1_tiny3.npz, g2a_tiny1.npz, g2b_cap.npz   full tensors of tiny nets
  g3_full.json                               full-size cfg-S (N=2) digest: sha256 of weights,
                                             probes of the output, per-layer norms, loss
  g4_ops.npz                                 per-op known-answer cases (pool ties/NaN, convT map,
                                             BN biased/unbiased var, masked-L1 edge cases, Adam)
  g5_init.json                               sha256 of default-initialised state_dicts (RNG order)
"""
import hashlib
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
if "torchvision" not in sys.modules:            # see module docstring
    tv = types.ModuleType("torchvision")
    tv.transforms = types.ModuleType("torchvision.transforms")
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.transforms"] = tv.transforms

from lib.UNet import UNet                                   # noqa: E402  (reference)
from lib.data_normalization import denormalize_torch        # noqa: E402  (reference)

torch.set_num_threads(8)


def ref_loss(criterion, y_pred, y, loss_mask, mean, std):
    # drives lib/Trainer.py:87-100 through the reference's denormalize_torch
    y_pred_metric = denormalize_torch(y_pred, mean, std)
    y_metric = denormalize_torch(y, mean, std)
    y_pred_metric[loss_mask == 0] = 0
    y_metric[loss_mask == 0] = 0
    loss = criterion(y_pred_metric, y_metric)
    loss = loss * loss_mask.numel() / loss_mask.sum()
    return loss


def make_batch(n, c, t, seed, nodata_frac=0.05, mean_scale=50.0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, c, t, t, generator=g)
    y = x[:, 0:1] + 0.3 * torch.randn(n, 1, t, t, generator=g)
    mask = torch.rand(n, 1, t, t, generator=g) > nodata_frac
    mean = torch.randn(n, generator=g, dtype=torch.float64) * mean_scale
    std = torch.rand(n, generator=g) * 2.0 + 1.0
    return {"input": x, "target": y, "loss_mask": mask, "dsm_mean": mean, "dsm_std": std}


def sd_to_np(sd, prefix):
    return {prefix + k: v.detach().cpu().numpy().copy() for k, v in sd.items()}


def sha_sd(sd):
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(np.ascontiguousarray(v.detach().cpu().numpy()).tobytes())
    return h.hexdigest()


def tiny_case(name, kwargs, n, t, seed_w, seed_x, adam_steps=3, lr=2e-4, wd=1e-5):
    torch.manual_seed(seed_w)
    model = UNet(**kwargs)
    out = {}
    out.update(sd_to_np(model.state_dict(), "init/"))
    batch = make_batch(n, kwargs["n_input_channels"], t, seed_x)
    for k, v in batch.items():
        out["batch/" + k] = v.numpy().copy()

    # eval-mode output with the initial running stats
    model.eval()
    with torch.no_grad():
        out["y_eval_init"] = model(batch["input"]).numpy()

    # pooling inputs / indices via hooks on the reference's MaxPool2d modules
    pooled = {}

    def mk_hook(i):
        def hook(mod, inp, outp):
            _, idx = torch.nn.functional.max_pool2d(inp[0].detach(), 2, 2, return_indices=True)
            pooled[i] = idx
        return hook
    handles = [model.encoder[i][-1].register_forward_hook(mk_hook(i)) for i in range(len(model.encoder))]

    # one training iteration (lib/Trainer.py:159-179, 212-222)
    criterion = torch.nn.L1Loss(reduction="mean")
    optimizer = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=wd)
    model.train()
    y_pred = model(batch["input"])
    for h in handles:
        h.remove()
    mean = torch.flatten(batch["dsm_mean"])
    std = torch.flatten(batch["dsm_std"])
    loss = ref_loss(criterion, y_pred, batch["target"], batch["loss_mask"], mean, std)
    loss.backward()
    out["y_train"] = y_pred.detach().numpy().copy()
    out["loss"] = np.float32(loss.item())
    for k, p in model.named_parameters():
        out["grad/" + k] = p.grad.detach().numpy().copy()
    for i, idx in pooled.items():
        out[f"poolidx/{i}"] = idx.numpy().astype(np.int32)
    out.update(sd_to_np({k: v for k, v in model.state_dict().items() if "running" in k or "num_batches" in k},
                        "bn_after1/"))
    optimizer.step()
    for p in model.parameters():
        p.grad = None
    out.update(sd_to_np(model.state_dict(), "after1/"))
    losses = [loss.item()]
    for _ in range(adam_steps - 1):
        y_pred = model(batch["input"])
        loss = ref_loss(criterion, y_pred, batch["target"], batch["loss_mask"], mean, std)
        loss.backward()
        optimizer.step()
        for p in model.parameters():
            p.grad = None
        losses.append(loss.item())
    out.update(sd_to_np(model.state_dict(), f"after{adam_steps}/"))
    out["losses"] = np.array(losses, dtype=np.float32)
    model.eval()
    with torch.no_grad():
        out[f"y_eval_after{adam_steps}"] = model(batch["input"]).numpy()
    out["kwargs_json"] = np.array(json.dumps(kwargs))
    out["meta_json"] = np.array(json.dumps({"n": n, "t": t, "seed_w": seed_w, "seed_x": seed_x,
                                            "adam_steps": adam_steps, "lr": lr, "wd": wd}))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "loss", losses, "bytes", os.path.getsize(os.path.join(HERE, name + ".npz")))


def full_digest():
    """G3: cfg-S architecture at N=2 -- digest only (weights alone are 50 MB)."""
    kwargs = dict(n_input_channels=3, start_kernel=64, depth=5, bias_conv_layer=True)
    torch.manual_seed(0)
    model = UNet(**kwargs)
    sha = sha_sd(model.state_dict())
    batch = make_batch(2, 3, 256, 1234, mean_scale=50.0)
    acts = {}

    def hook(name):
        def f(mod, inp, outp):
            acts[name] = float(outp.detach().double().pow(2).sum().sqrt())
        return f
    hs = []
    for name, mod in model.named_modules():
        if isinstance(mod, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
            hs.append(mod.register_forward_hook(hook(name)))
    model.train()
    y_pred = model(batch["input"])
    for h in hs:
        h.remove()
    criterion = torch.nn.L1Loss(reduction="mean")
    loss = ref_loss(criterion, y_pred, batch["target"], batch["loss_mask"],
                    torch.flatten(batch["dsm_mean"]), torch.flatten(batch["dsm_std"]))
    loss.backward()
    g = torch.Generator().manual_seed(7)
    coords = torch.stack([torch.randint(0, 2, (64,), generator=g), torch.randint(0, 256, (64,), generator=g),
                          torch.randint(0, 256, (64,), generator=g)], 1)
    probes = [float(y_pred[int(n), 0, int(y), int(x)]) for n, y, x in coords]
    grad_norms = {k: float(p.grad.double().pow(2).sum().sqrt()) for k, p in model.named_parameters()}
    bn = {k: [float(v.double().mean()), float(v.double().pow(2).sum().sqrt())]
          for k, v in model.state_dict().items() if "running" in k}
    dig = {"kwargs": kwargs, "seed_w": 0, "batch": {"n": 2, "c": 3, "t": 256, "seed": 1234, "mean_scale": 50.0},
           "state_sha256": sha, "loss": float(loss.item()), "coords": coords.tolist(), "probes": probes,
           "act_l2": acts, "grad_l2": grad_norms, "bn_running": bn,
           "y_l2": float(y_pred.detach().double().pow(2).sum().sqrt())}
    with open(os.path.join(HERE, "g3_full.json"), "w") as f:
        json.dump(dig, f, indent=1)
    print("g3 loss", dig["loss"])


def op_cases():
    """G4: per-op known answers computed with the torch ops the reference calls."""
    F = torch.nn.functional
    out = {}
    # max-pool ties / NaN (nn.MaxPool2d(2,2), lib/UNet.py:161,167)
    x = torch.tensor([[1., 1., 0., 0., 2., 3., float("nan"), 1.],
                      [1., 1., 0., 0., 3., 2., 5., float("nan")],
                      [0., -1., 4., 4., float("nan"), float("nan"), -0., 0.],
                      [-1., 0., 4., 5., 1., float("nan"), 0., -0.]]).view(1, 1, 4, 8)
    p, idx = F.max_pool2d(x, 2, 2, return_indices=True)
    out["pool/x"], out["pool/y"], out["pool/idx"] = x.numpy(), p.numpy(), idx.numpy().astype(np.int32)
    g = torch.Generator().manual_seed(3)
    xr = torch.relu(torch.randn(2, 5, 8, 8, generator=g))
    xr.requires_grad_(True)
    p, idx = F.max_pool2d(xr, 2, 2, return_indices=True)
    gy = torch.randn(p.shape, generator=g)
    p.backward(gy)
    out["pool2/x"], out["pool2/y"], out["pool2/idx"] = xr.detach().numpy(), p.detach().numpy(), idx.numpy().astype(np.int32)
    out["pool2/gy"], out["pool2/gx"] = gy.numpy(), xr.grad.numpy()
    # transposed conv k2 s2 (+bias) index mapping (lib/UNet.py:21)
    x = torch.randn(2, 4, 4, 8, generator=g, requires_grad=True)
    w = torch.randn(4, 6, 2, 2, generator=g, requires_grad=True)
    b = torch.randn(6, generator=g, requires_grad=True)
    y = F.conv_transpose2d(x, w, b, stride=2)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    for k, v in dict(x=x, w=w, b=b, y=y, gy=gy, gx=x.grad, gw=w.grad, gb=b.grad).items():
        out["convt/" + k] = v.detach().numpy()
    # conv3x3 pad 1 (lib/UNet.py:4-5)
    x = torch.randn(2, 4, 8, 4, generator=g, requires_grad=True)
    w = torch.randn(8, 4, 3, 3, generator=g, requires_grad=True)
    y = F.conv2d(x, w, None, 1, 1)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    for k, v in dict(x=x, w=w, y=y, gy=gy, gx=x.grad, gw=w.grad).items():
        out["conv/" + k] = v.detach().numpy()
    # BatchNorm2d train: biased var in the output, unbiased in running_var (lib/UNet.py:45)
    x = (torch.randn(3, 4, 5, 5, generator=g) * 2.0 + 1.5).requires_grad_(True)
    gam = torch.randn(4, generator=g).requires_grad_(True)
    bet = torch.randn(4, generator=g).requires_grad_(True)
    rm, rv = torch.zeros(4), torch.ones(4)
    y = F.batch_norm(x, rm, rv, gam, bet, True, 0.1, 1e-5)
    r = torch.relu(y)
    gy = torch.randn(y.shape, generator=g)
    r.backward(gy)
    for k, v in dict(x=x, gamma=gam, beta=bet, y=r, gy=gy, gx=x.grad, ggamma=gam.grad, gbeta=bet.grad,
                     running_mean=rm, running_var=rv).items():
        out["bn/" + k] = v.detach().numpy()
    # masked L1: one fully masked sample, and the all-masked batch (0/0 -> nan) (lib/Trainer.py:87-100)
    crit = torch.nn.L1Loss(reduction="mean")
    yp = torch.randn(3, 1, 4, 4, generator=g, requires_grad=True)
    yt = torch.randn(3, 1, 4, 4, generator=g)
    m = torch.rand(3, 1, 4, 4, generator=g) > 0.3
    m[1] = False
    mean = torch.tensor([412.25, -3.5, 1000.125], dtype=torch.float64)
    std = torch.tensor([3.0, 1.5, 0.25])
    loss = ref_loss(crit, yp, yt, m, mean, std)
    loss.backward()
    for k, v in dict(yp=yp, yt=yt, mask=m, mean=mean, std=std, loss=loss, gyp=yp.grad).items():
        out["l1/" + k] = v.detach().numpy()
    yp2 = yp.detach().clone().requires_grad_(True)
    loss0 = ref_loss(crit, yp2, yt, torch.zeros_like(m), mean, std)
    loss0.backward()
    out["l1/loss_allmasked"] = loss0.detach().numpy()
    out["l1/gyp_allmasked"] = yp2.grad.numpy()
    # Adam: step 1 and step 1000 bias correction (lib/utils.py:329-331)
    p = torch.randn(257, generator=g).requires_grad_(True)
    opt = torch.optim.Adam([p], lr=2e-4, weight_decay=1e-5)
    out["adam/p0"] = p.detach().numpy().copy()
    gs = torch.randn(3, 257, generator=g)
    out["adam/g"] = gs.numpy()
    p.grad = gs[0].clone()
    opt.step()
    out["adam/p1"] = p.detach().numpy().copy()
    st = opt.state[p]
    st["step"] = torch.tensor(999.0)
    p.grad = gs[1].clone()
    opt.step()
    out["adam/p1000"] = p.detach().numpy().copy()
    out["adam/m1000"] = st["exp_avg"].numpy().copy()
    out["adam/v1000"] = st["exp_avg_sq"].numpy().copy()
    np.savez_compressed(os.path.join(HERE, "g4_ops.npz"), **out)
    print("g4 ok")


def init_digest():
    dig = {}
    for seed in (0, 1):
        for (c, d) in ((3, 5), (1, 5), (2, 6)):
            torch.manual_seed(seed)
            m = UNet(n_input_channels=c, start_kernel=64, depth=d, bias_conv_layer=True)
            sd = m.state_dict()
            dig[f"seed{seed}_c{c}_d{d}"] = {"sha256": sha_sd(sd), "n_entries": len(sd),
                                            "n_params": sum(p.numel() for p in m.parameters()),
                                            "keys_sha256": hashlib.sha256("\n".join(sd.keys()).encode()).hexdigest()}
    with open(os.path.join(HERE, "g5_init.json"), "w") as f:
        json.dump(dig, f, indent=1)
    print("g5", {k: v["n_params"] for k, v in dig.items()})


if __name__ == "__main__":
    tiny_case("g1_tiny3", dict(n_input_channels=3, start_kernel=8, depth=3, bias_conv_layer=True), n=2, t=32,
              seed_w=0, seed_x=1)
    tiny_case("g2a_tiny1", dict(n_input_channels=1, start_kernel=4, depth=2, bias_conv_layer=False), n=3, t=16,
              seed_w=5, seed_x=6)
    tiny_case("g2b_cap", dict(n_input_channels=2, start_kernel=16, depth=4, max_filter_depth=32,
                              bias_conv_layer=True, outer_skip=False), n=1, t=64, seed_w=7, seed_x=8)
    tiny_case("g7_nobn", dict(n_input_channels=2, start_kernel=8, depth=2, do_BN=False, bias_conv_layer=True), n=2,
              t=16, seed_w=11, seed_x=12)
    tiny_case("g8_lrelu_oskipbn", dict(n_input_channels=3, start_kernel=8, depth=2, act_fn_encoder="lrelu",
                                       act_fn_decoder="lrelu", act_fn_bottleneck="lrelu", outer_skip_BN=True,
                                       bias_conv_layer=True), n=3, t=16, seed_w=13, seed_x=14)
    tiny_case("g11_prelu", dict(n_input_channels=2, start_kernel=8, depth=2, act_fn_encoder="prelu",
                                act_fn_decoder="prelu", act_fn_bottleneck="prelu", bias_conv_layer=True), n=2, t=16,
              seed_w=15, seed_x=16, lr=5e-3)
    tiny_case("g12_bilinear", dict(n_input_channels=3, start_kernel=8, depth=3, up_mode="bilinear",
                                   bias_conv_layer=True), n=2, t=32, seed_w=17, seed_x=18, lr=2e-3)
    op_cases()
    init_digest()
    full_digest()
