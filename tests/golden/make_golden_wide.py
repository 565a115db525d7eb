#!/usr/bin/env python3
"""Golden fixtures for the widened constructor domain, from the REFERENCE implementation.

Run:  python tests/golden/make_golden_wide.py      (build container only: needs /root/reference)

Same key layout as make_golden.tiny_case (so the tiny-net tests read them unchanged) plus
  grad_input        d loss / d input of the first training iteration (input.requires_grad_())
  opt_json          {"name": "adam" | "sgd", "lr", "wd", "momentum", "nesterov"}

  g13_sk6.npz       start_kernel = 6 (channel counts 6, 12: not multiples of 4), lib/UNet.py:105-107
  g14_cin9.npz      n_input_channels = 9 (> 6, not a multiple of 4)
  g15_cin8.npz      n_input_channels = 8 (> 6: first convolution on the generic kernels), PReLU, outer-skip BN
  g16_sgd.npz       torch.optim.SGD(lr, weight_decay) as lib/utils.py:332-334 builds it
  g17_sgd_mom.npz   torch.optim.SGD with momentum 0.9 + nesterov (torch's options, not used by the reference's configs)
"""
import json
import os

import numpy as np
import torch

from make_golden import HERE, UNet, make_batch, ref_loss, sd_to_np


def wide_case(name, kwargs, n, t, seed_w, seed_x, opt, steps=3):
    torch.manual_seed(seed_w)
    model = UNet(**kwargs)
    out = {}
    out.update(sd_to_np(model.state_dict(), "init/"))
    batch = make_batch(n, kwargs["n_input_channels"], t, seed_x)
    for k, v in batch.items():
        out["batch/" + k] = v.numpy().copy()
    model.eval()
    with torch.no_grad():
        out["y_eval_init"] = model(batch["input"]).numpy()
    criterion = torch.nn.L1Loss(reduction="mean")
    if opt["name"] == "adam":
        optimizer = torch.optim.Adam(model.parameters(), lr=opt["lr"], weight_decay=opt["wd"])
    else:
        optimizer = torch.optim.SGD(model.parameters(), lr=opt["lr"], weight_decay=opt["wd"], momentum=opt["momentum"],
                                    nesterov=opt["nesterov"])
    model.train()
    mean, std = torch.flatten(batch["dsm_mean"]), torch.flatten(batch["dsm_std"])
    x = batch["input"].clone().requires_grad_(True)
    y_pred = model(x)
    loss = ref_loss(criterion, y_pred, batch["target"], batch["loss_mask"], mean, std)
    loss.backward()
    out["y_train"] = y_pred.detach().numpy().copy()
    out["loss"] = np.float32(loss.item())
    out["grad_input"] = x.grad.numpy().copy()
    for k, p in model.named_parameters():
        out["grad/" + k] = p.grad.detach().numpy().copy()
    out.update(sd_to_np({k: v for k, v in model.state_dict().items() if "running" in k or "num_batches" in k}, "bn_after1/"))
    optimizer.step()
    for p in model.parameters():
        p.grad = None
    out.update(sd_to_np(model.state_dict(), "after1/"))
    losses = [loss.item()]
    for _ in range(steps - 1):
        y_pred = model(batch["input"])
        loss = ref_loss(criterion, y_pred, batch["target"], batch["loss_mask"], mean, std)
        loss.backward()
        optimizer.step()
        for p in model.parameters():
            p.grad = None
        losses.append(loss.item())
    out.update(sd_to_np(model.state_dict(), f"after{steps}/"))
    out["losses"] = np.array(losses, dtype=np.float32)
    model.eval()
    with torch.no_grad():
        out[f"y_eval_after{steps}"] = model(batch["input"]).numpy()
    out["kwargs_json"] = np.array(json.dumps(kwargs))
    out["meta_json"] = np.array(json.dumps({"n": n, "t": t, "seed_w": seed_w, "seed_x": seed_x, "adam_steps": steps,
                                            "lr": opt["lr"], "wd": opt["wd"]}))
    out["opt_json"] = np.array(json.dumps(opt))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "loss", losses, "bytes", os.path.getsize(os.path.join(HERE, name + ".npz")))


ADAM = {"name": "adam", "lr": 2e-4, "wd": 1e-5, "momentum": 0.0, "nesterov": False}

if __name__ == "__main__":
    wide_case("g13_sk6", dict(n_input_channels=3, start_kernel=6, depth=2, bias_conv_layer=True), n=2, t=16, seed_w=21,
              seed_x=22, opt=ADAM)
    wide_case("g14_cin9", dict(n_input_channels=9, start_kernel=8, depth=2, bias_conv_layer=True), n=2, t=16, seed_w=23,
              seed_x=24, opt=ADAM)
    wide_case("g15_cin8", dict(n_input_channels=8, start_kernel=8, depth=2, act_fn_encoder="prelu", act_fn_decoder="prelu",
                               act_fn_bottleneck="prelu", outer_skip_BN=True, bias_conv_layer=True), n=3, t=16, seed_w=25,
              seed_x=26, opt=dict(ADAM, lr=2e-3))
    wide_case("g16_sgd", dict(n_input_channels=3, start_kernel=8, depth=2, bias_conv_layer=True), n=2, t=16, seed_w=27,
              seed_x=28, opt={"name": "sgd", "lr": 1e-2, "wd": 1e-3, "momentum": 0.0, "nesterov": False})
    wide_case("g17_sgd_mom", dict(n_input_channels=2, start_kernel=8, depth=2, bias_conv_layer=False), n=2, t=16, seed_w=29,
              seed_x=30, opt={"name": "sgd", "lr": 1e-2, "wd": 1e-3, "momentum": 0.9, "nesterov": True})
