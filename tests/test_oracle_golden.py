"""The oracle (oracle/unet_oracle.py) must reproduce fixtures generated from the REFERENCE
(tests/golden/make_golden.py imported lib/UNet.py).  CPU only; this is what pins parity."""
import hashlib
import json

import numpy as np
import pytest
import torch

from conftest import load_json, load_npz
from oracle import unet_oracle as O

TINY = ["g1_tiny3.npz", "g2a_tiny1.npz", "g2b_cap.npz", "g7_nobn.npz", "g8_lrelu_oskipbn.npz", "g11_prelu.npz", "g12_bilinear.npz"]
# widened constructor domain + SGD (tests/golden/make_golden_wide.py): same keys + grad_input + opt_json, no pool indices
WIDE = ["g13_sk6.npz", "g14_cin9.npz", "g15_cin8.npz", "g16_sgd.npz", "g17_sgd_mom.npz"]


def _spec(kwargs):
    return O.Spec(**{"depth": 8, **kwargs})


def _sub(g, prefix):
    return {k[len(prefix):]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith(prefix)}


def _sha_sd(sd):
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(np.ascontiguousarray(v.detach().cpu().numpy()).tobytes())
    return h.hexdigest()


@pytest.mark.parametrize("name", TINY + WIDE)
def test_tiny_net_matches_reference(name):
    g = load_npz(name)
    kwargs = json.loads(str(g["kwargs_json"]))
    meta = json.loads(str(g["meta_json"]))
    opt = json.loads(str(g["opt_json"])) if "opt_json" in g else {"name": "adam"}
    sgd = {"momentum": opt["momentum"], "nesterov": opt["nesterov"]} if opt["name"] == "sgd" else None
    spec = _spec(kwargs)
    sd = _sub(g, "init/")
    assert list(sd.keys()) == [k for k, _, _ in O.param_layout(spec)]
    for k, shape, _ in O.param_layout(spec):
        assert tuple(sd[k].shape) == tuple(shape), k
    # same seed -> same weights (RNG draw order)
    sd_init = O.init_state_dict(spec, meta["seed_w"])
    for k in sd:
        assert torch.equal(sd_init[k], sd[k]), k
    batch = _sub(g, "batch/")
    # eval forward with initial running stats
    y = O.forward(sd, batch["input"], spec, training=False)
    np.testing.assert_allclose(y.numpy(), g["y_eval_init"], rtol=0, atol=1e-6)
    # training iterations
    state = {}
    keep = {}
    losses = []
    for it in range(meta["adam_steps"]):
        loss, grads = O.train_step(sd, batch, spec, state, lr=meta["lr"], weight_decay=meta["wd"],
                                   keep=keep if it == 0 else None, sgd=sgd)
        losses.append(loss)
        if it == 0:
            np.testing.assert_allclose(keep["y_pred"].numpy(), g["y_train"], rtol=0, atol=1e-6)
            assert abs(loss - float(g["loss"])) <= 1e-6 * abs(float(g["loss"]))
            for k, gr in grads.items():
                ref = g["grad/" + k]
                num = np.linalg.norm(gr.numpy().astype(np.float64) - ref)
                den = np.linalg.norm(ref.astype(np.float64)) + 1e-30
                assert num / den < 1e-5, (k, num / den)
            for i in range(spec.depth):
                if f"poolidx/{i}" in g:
                    assert np.array_equal(keep[f"idx{i}"].numpy().astype(np.int32), g[f"poolidx/{i}"]), i
            if "grad_input" in g:
                ref = g["grad_input"].astype(np.float64)
                num = np.linalg.norm(keep["grad_input"].numpy().astype(np.float64) - ref)
                assert num / (np.linalg.norm(ref) + 1e-30) < 1e-5
            for k, v in _sub(g, "bn_after1/").items():
                np.testing.assert_allclose(sd[k].numpy(), v.numpy(), rtol=1e-6, atol=1e-7, err_msg=k)
            for k, v in _sub(g, "after1/").items():
                np.testing.assert_allclose(sd[k].numpy(), v.numpy(), rtol=1e-5, atol=1e-7, err_msg=k)
    np.testing.assert_allclose(np.array(losses, np.float32), g["losses"], rtol=2e-6)
    for k, v in _sub(g, f"after{meta['adam_steps']}/").items():
        np.testing.assert_allclose(sd[k].numpy(), v.numpy(), rtol=1e-4, atol=1e-6, err_msg=k)
    y = O.forward(sd, batch["input"], spec, training=False)
    np.testing.assert_allclose(y.numpy(), g[f"y_eval_after{meta['adam_steps']}"], rtol=0, atol=1e-5)


def test_init_order_digests():
    dig = load_json("g5_init.json")
    for name, d in dig.items():
        seed = int(name.split("_")[0][4:])
        c = int(name.split("_")[1][1:])
        depth = int(name.split("_")[2][1:])
        if depth == 6 and seed == 1:
            continue                      # keep CPU suite short; (seed0, d6) covers the shape
        spec = O.Spec(n_input_channels=c, start_kernel=64, depth=depth, bias_conv_layer=True)
        sd = O.init_state_dict(spec, seed)
        assert len(sd) == d["n_entries"]
        assert hashlib.sha256("\n".join(sd.keys()).encode()).hexdigest() == d["keys_sha256"]
        assert sum(v.numel() for k, v in sd.items() if k in O.param_keys(spec)) == d["n_params"]
        assert _sha_sd(sd) == d["sha256"], name


def test_full_size_digest():
    """G3: cfg-S architecture, N=2 -- oracle vs the reference's digest."""
    d = load_json("g3_full.json")
    spec = _spec(d["kwargs"])
    sd = O.init_state_dict(spec, d["seed_w"])
    assert _sha_sd(sd) == d["state_sha256"]
    b = d["batch"]
    g = torch.Generator().manual_seed(b["seed"])
    x = torch.randn(b["n"], b["c"], b["t"], b["t"], generator=g)
    y = x[:, 0:1] + 0.3 * torch.randn(b["n"], 1, b["t"], b["t"], generator=g)
    mask = torch.rand(b["n"], 1, b["t"], b["t"], generator=g) > 0.05
    mean = torch.randn(b["n"], generator=g, dtype=torch.float64) * b["mean_scale"]
    std = torch.rand(b["n"], generator=g) * 2.0 + 1.0
    leaves = {k: sd[k].clone().requires_grad_(True) for k in O.param_keys(spec)}
    work = dict(sd)
    work.update(leaves)
    keep = {}
    yp = O.forward(work, x, spec, training=True, keep=keep)
    loss = O.masked_l1_loss(yp, y, mask, mean, std)
    assert abs(float(loss) - d["loss"]) < 2e-6 * abs(d["loss"])
    for (n, yy, xx), p in zip(d["coords"], d["probes"]):
        assert abs(float(yp[n, 0, yy, xx]) - p) < 2e-5
    assert abs(float(yp.detach().double().pow(2).sum().sqrt()) - d["y_l2"]) < 1e-5 * d["y_l2"]
    names = {"encoder.0.0.0": "z0", "bottleneck.0": "zb", "decoder.0.0": "u0", "decoder.4": "u4",
             "decoder.3.1.0": "zd3", "last_layer": "res"}
    for mod, key in names.items():
        ref = d["act_l2"][mod]
        got = float(keep[key].detach().double().pow(2).sum().sqrt())
        assert abs(got - ref) < 1e-5 * ref, mod
    grads = torch.autograd.grad(loss, list(leaves.values()))
    for k, gr in zip(leaves, grads):
        ref = d["grad_l2"][k]
        got = float(gr.double().pow(2).sum().sqrt())
        assert abs(got - ref) <= 1e-3 * ref + 1e-12, (k, got, ref)


def test_ops_known_answers(g4):
    F = torch.nn.functional
    t = lambda k: torch.from_numpy(g4[k].copy())
    # pooling ties / NaN
    p, idx = F.max_pool2d(t("pool/x"), 2, 2, return_indices=True)
    assert np.array_equal(idx.numpy().astype(np.int32), g4["pool/idx"])
    np.testing.assert_array_equal(p.numpy(), g4["pool/y"])
    # masked L1 incl. a fully masked sample and the all-masked batch
    yp = t("l1/yp").requires_grad_(True)
    loss = O.masked_l1_loss(yp, t("l1/yt"), t("l1/mask"), t("l1/mean"), t("l1/std"))
    loss.backward()
    np.testing.assert_allclose(loss.detach().numpy(), g4["l1/loss"], rtol=1e-6)
    np.testing.assert_allclose(yp.grad.numpy(), g4["l1/gyp"], rtol=1e-6, atol=0)
    loss0 = O.masked_l1_loss(t("l1/yp"), t("l1/yt"), torch.zeros_like(t("l1/mask")), t("l1/mean"), t("l1/std"))
    assert np.isnan(float(loss0)) and np.isnan(float(g4["l1/loss_allmasked"]))
    # Adam step 1 / step 1000
    p = t("adam/p0")
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    gs = t("adam/g")
    O.adam_step([p], [gs[0]], [m], [v], 1)
    np.testing.assert_allclose(p.numpy(), g4["adam/p1"], rtol=1e-6, atol=1e-8)
    O.adam_step([p], [gs[1]], [m], [v], 1000)
    np.testing.assert_allclose(p.numpy(), g4["adam/p1000"], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(m.numpy(), g4["adam/m1000"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(v.numpy(), g4["adam/v1000"], rtol=1e-6, atol=1e-12)
