"""GPU sample assembly (-m gpu) against samples produced by the reference's DsmOrthoDataset.__getitem__ (g9 fixture),
and as a drop-in producer for the training step."""
import numpy as np
import pytest
import torch

from conftest import load_npz

pytestmark = pytest.mark.gpu


def test_gpu_sampler_matches_reference_samples():
    from resdepth_amd import GpuPatchSampler
    g = load_npz("g9_samples.npz")
    t = int(g["tile"])
    orth = np.ascontiguousarray(g["orthos"].transpose(2, 0, 1))
    smp = GpuPatchSampler(g["dsm_in"], g["dsm_gt"], orth, tile_size=t, nodata=float(g["nodata"]), dsm_std=float(g["dsm_std"]),
                          ortho_mean=float(g["ortho_mean"]), ortho_std=float(g["ortho_std"]))
    b = smp.sample(g["pos"], g["pairs"], g["aug"])
    for i in range(len(g["pos"])):
        assert np.array_equal(b["loss_mask"][i].cpu().numpy(), g[f"s{i}/loss_mask"]), i
        # the per-patch mean is a float32 sum in numpy and an fp64 sum here: values agree to 1 ulp of the mean / std
        np.testing.assert_allclose(b["input"][i].cpu().numpy(), g[f"s{i}/input"], rtol=0, atol=3e-5)
        np.testing.assert_allclose(b["target"][i].cpu().numpy(), g[f"s{i}/target"], rtol=0, atol=3e-5)
        assert abs(float(b["dsm_mean"][i]) - float(g[f"s{i}/dsm_mean"])) <= 1e-6 * abs(float(g[f"s{i}/dsm_mean"]))
    smp2 = GpuPatchSampler(g["dsm_in"], g["dsm_gt"], orth, tile_size=t, nodata=float(g["nodata"]), dsm_std=float(g["dsm_std"]),
                           ortho_mean=None, ortho_std=float(g["ortho_std"]))
    b2 = smp2.sample(g["pos"][:3], g["pairs"][:3], None)
    for i in range(3):
        assert np.array_equal(b2["loss_mask"][i].cpu().numpy(), g[f"n{i}/loss_mask"])
        np.testing.assert_allclose(b2["input"][i].cpu().numpy(), g[f"n{i}/input"], rtol=0, atol=3e-5)
    # ortho channels with a GIVEN mean involve no reduction: bit-exact
    for i in range(len(g["pos"])):
        assert np.array_equal(b["input"][i, 1:].cpu().numpy(), g[f"s{i}/input"][1:]), i


def test_sampler_feeds_the_training_step():
    from resdepth_amd import GpuPatchSampler, UNet, FusedAdam, masked_l1_loss
    g = torch.Generator().manual_seed(0)
    H = W = 512
    dsm = torch.randn(H, W, generator=g) * 3 + 400
    gt = dsm + torch.randn(H, W, generator=g)
    orth = torch.rand(3, H, W, generator=g) * 200
    smp = GpuPatchSampler(dsm, gt, orth, tile_size=64, dsm_std=3.0, ortho_mean=100.0, ortho_std=50.0)
    model = UNet(n_input_channels=3, start_kernel=8, depth=3, bias_conv_layer=True).to("cuda:0").train()
    opt = FusedAdam(model.parameters(), lr=1e-3)
    losses = []
    for it in range(6):
        b = smp.random_batch(8, [0, 2], generator=g)
        assert b["input"].shape == (8, 3, 64, 64) and b["loss_mask"].dtype == torch.bool
        loss = masked_l1_loss(model(b["input"]), b["target"], b["loss_mask"], b["dsm_mean"], b["dsm_std"])
        loss.backward()
        opt.step()
        for p in model.parameters():
            p.grad = None
        losses.append(float(loss))
    assert np.isfinite(losses).all() and losses[-1] < losses[0]


def test_stream_batches_equals_consecutive_random_batches():
    """GpuPatchSampler.stream_batches assembles batch k + 1 on a side stream while batch k is consumed; the batches (and a short
    training run on them) are the ones consecutive random_batch calls give, bit for bit, at prefetch depth 0, 1 and 3."""
    from resdepth_amd import GpuPatchSampler, UNet, FusedAdam, masked_l1_loss
    g0 = torch.Generator().manual_seed(3)
    H = W = 512
    dsm = torch.randn(H, W, generator=g0) * 3 + 400
    gt = dsm + torch.randn(H, W, generator=g0)
    orth = torch.rand(3, H, W, generator=g0) * 200
    smp = GpuPatchSampler(dsm, gt, orth, tile_size=64, dsm_std=3.0, ortho_mean=None, ortho_std=50.0)

    def run(depth):
        g = torch.Generator().manual_seed(11)
        torch.manual_seed(0)
        model = UNet(n_input_channels=3, start_kernel=8, depth=3, bias_conv_layer=True).to("cuda:0").train()
        opt = FusedAdam(model.parameters(), lr=1e-3)
        it = (smp.random_batch(8, [1, 2], generator=g) for _ in range(7)) if depth is None else \
            smp.stream_batches(7, 8, [1, 2], generator=g, prefetch=depth)
        seen, losses = [], []
        for b in it:
            seen.append({k: v.clone() for k, v in b.items()})
            loss = masked_l1_loss(model(b["input"]), b["target"], b["loss_mask"], b["dsm_mean"], b["dsm_std"])
            loss.backward()
            opt.step()
            for p in model.parameters():
                p.grad = None
            losses.append(loss.detach())
        torch.cuda.synchronize()
        return seen, [float(v) for v in losses], {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}

    ref = run(None)
    assert len(ref[0]) == 7
    for depth in (0, 1, 3):
        got = run(depth)
        assert len(got[0]) == 7 and got[1] == ref[1], depth
        for a, b in zip(got[0], ref[0]):
            for k in b:
                assert torch.equal(a[k], b[k]), (depth, k)
        for k, v in ref[2].items():
            assert torch.equal(got[2][k], v), (depth, k)
