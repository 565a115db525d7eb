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


def test_sampler_loader_drives_the_trainer_shell(tmp_path):
    """SamplerLoader as the Trainer's `trainloader` (lib/Trainer.py:61-64,159-222 consume any iterable of batch dicts with a
    length): two epochs of GPU-assembled batches through resdepth_amd.Trainer give the losses and weights of the same batches fed
    by hand, bit for bit."""
    import types
    from resdepth_amd import GpuPatchSampler, SamplerLoader, UNet, FusedAdam, Trainer, masked_l1_loss
    g0 = torch.Generator().manual_seed(5)
    H = W = 384
    dsm = torch.randn(H, W, generator=g0) * 3 + 400
    smp = GpuPatchSampler(dsm, dsm + torch.randn(H, W, generator=g0), torch.rand(2, H, W, generator=g0) * 200, tile_size=64,
                          dsm_std=3.0, ortho_mean=100.0, ortho_std=50.0)
    kw = dict(n_input_channels=3, start_kernel=8, depth=3, bias_conv_layer=True)

    torch.manual_seed(0)
    model = UNet(**kw)
    opt = FusedAdam(model.parameters(), lr=1e-3)
    loader = SamplerLoader(smp, 5, 4, [0, 1], generator=torch.Generator().manual_seed(21))
    args = types.SimpleNamespace(model=model, optimizer=opt, scheduler=None, criterion=torch.nn.L1Loss(reduction="mean"),
                                 trainloader=loader, valloader=None, n_epochs=2, evaluate_rate=10, save_model_rate=10 ** 9,
                                 freq_average_train_loss=10 ** 9, save_dir=str(tmp_path), log_file=None,
                                 checkpoint_dir=str(tmp_path / "ck"), tboard_log_dir=str(tmp_path / "tb"), pretrained_path=None)
    tr = Trainer(args)
    assert tr.batch_size == 4 and len(loader) == 5
    meters = [tr.inference_one_epoch(e, "train")["MAE_metric"].avg for e in range(2)]
    torch.cuda.synchronize()

    # by hand: the Trainer's constructor peeked at the first batch (lib/Trainer.py:61-64) -- with prefetch 1 the abandoned iterator
    # had drawn two --, then two epochs of five
    torch.manual_seed(0)
    ref = UNet(**kw).to("cuda:0").train()
    ropt = FusedAdam(ref.parameters(), lr=1e-3)
    g = torch.Generator().manual_seed(21)
    for _ in range(2):
        smp.random_batch(4, [0, 1], generator=g)
    want = []
    for e in range(2):
        acc = []
        for _ in range(5):
            b = smp.random_batch(4, [0, 1], generator=g)
            loss = masked_l1_loss(ref(b["input"]), b["target"], b["loss_mask"], b["dsm_mean"], b["dsm_std"])
            loss.backward()
            ropt.step()
            for p in ref.parameters():
                p.grad = None
            acc.append(float(loss))
        want.append(sum(acc) / len(acc))
    assert meters == pytest.approx(want, rel=1e-12)
    for (k, a), (_, b) in zip(model.state_dict().items(), ref.state_dict().items()):
        assert torch.equal(a, b), k
