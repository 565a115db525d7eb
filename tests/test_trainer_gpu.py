"""Trainer surface on the GPU (-m gpu): reference-shaped args object, DataLoader-collated batch dicts, the
reference's checkpoint dict, resume, and equality of the logged metric with the oracle's loop."""
import os
import types

import pytest
import torch
from torch.utils.data import DataLoader

from oracle import unet_oracle as O

pytestmark = pytest.mark.gpu


def _args(tmp, model, opt, sched, train, val, n_epochs, pretrained=None):
    return types.SimpleNamespace(
        model=model, optimizer=opt, scheduler=sched, criterion=torch.nn.L1Loss(reduction="mean"),
        trainloader=train, valloader=val, n_epochs=n_epochs, evaluate_rate=1, save_model_rate=1,
        freq_average_train_loss=20, save_dir=str(tmp), log_file=os.path.join(str(tmp), "training.log"),
        checkpoint_dir=os.path.join(str(tmp), "checkpoints"), tboard_log_dir=os.path.join(str(tmp), "tb"),
        pretrained_path=pretrained)


def test_trainer_train_checkpoint_resume_and_metric_parity(tmp_path):
    from resdepth_amd import UNet, FusedAdam, Trainer, SyntheticDsmOrthoDataset
    kw = dict(n_input_channels=2, start_kernel=8, depth=2, bias_conv_layer=True)
    spec = O.Spec(**kw)
    ds_train = SyntheticDsmOrthoDataset(8, 2, 32, seed=1)
    ds_val = SyntheticDsmOrthoDataset(4, 2, 32, seed=2)
    train = DataLoader(ds_train, batch_size=4, shuffle=False)
    val = DataLoader(ds_val, batch_size=4, shuffle=False)
    torch.manual_seed(0)
    model = UNet(**kw)
    sd_ref = {k: v.clone() for k, v in model.state_dict().items()}
    opt = FusedAdam(model.parameters(), lr=1e-3, weight_decay=1e-5)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=1, gamma=0.5)
    tr = Trainer(_args(tmp_path, model, opt, sched, train, val, n_epochs=2))
    assert tr.batch_size == 4 and tr.hparams["optimizer"] == "FusedAdam" and tr.hparams["step_size"] == 1
    stats = tr.inference_one_batch(next(iter(val)), "val")
    assert set(stats) == {"MAE_metric"} and isinstance(stats["MAE_metric"], float)
    tr.train()
    ck = os.path.join(str(tmp_path), "checkpoints")
    assert sorted(os.listdir(ck)) == ["Model_best.pth", "Model_last.pth"]
    last = torch.load(os.path.join(ck, "Model_last.pth"), weights_only=False)
    assert set(last) == {"epoch", "model_state_dict", "optimizer_state_dict", "loss_train", "loss_val",
                         "scheduler_state_dict"}
    assert last["epoch"] == 1 and list(last["model_state_dict"]) == list(sd_ref)
    assert abs(opt.param_groups[0]["lr"] - 0.25e-3) < 1e-12          # StepLR stepped after each validated epoch
    # oracle loop on the same data: train-loss of the last epoch and weights must agree
    state, losses = {}, []
    for ep in range(2):
        ep_losses = []
        for b in train:
            loss, _ = O.train_step(sd_ref, b, spec, state, lr=1e-3 * 0.5 ** ep, weight_decay=1e-5)
            ep_losses.append(loss)
        losses.append(sum(ep_losses) / len(ep_losses))
    assert abs(last["loss_train"] - losses[-1]) <= 1e-4 * abs(losses[-1])
    for k, v in last["model_state_dict"].items():
        if v.dtype.is_floating_point and "running" not in k:
            r = float((v.cpu().double() - sd_ref[k].double()).norm() / (sd_ref[k].double().norm() + 1e-30))
            assert r <= 5e-4, (k, r)
    # resume: epochs continue, optimizer / scheduler state restored
    torch.manual_seed(0)
    model2 = UNet(**kw)
    opt2 = FusedAdam(model2.parameters(), lr=1e-3, weight_decay=1e-5)
    sched2 = torch.optim.lr_scheduler.StepLR(opt2, step_size=1, gamma=0.5)
    tr2 = Trainer(_args(tmp_path / "resume", model2, opt2, sched2, train, val, n_epochs=1,
                        pretrained=os.path.join(ck, "Model_last.pth")))
    assert tr2.start_epoch == 2 and tr2.n_epochs == 3
    assert abs(opt2.param_groups[0]["lr"] - 0.25e-3) < 1e-12
    p0 = next(iter(opt2.state.values()))
    assert float(p0["step"]) == 4.0                                   # 2 epochs x 2 iterations
    tr2.train()
    assert os.path.isfile(os.path.join(str(tmp_path / "resume"), "checkpoints", "Model_last.pth"))


def test_h2d_prefetch_on_a_copy_stream_changes_no_bit(tmp_path):
    """DevicePrefetcher (batch k + 1 staged on a copy stream while batch k computes) against the copies on the compute stream in
    front of the forward (prefetch_batches=0, the reference's order, lib/Trainer.py:165-168): same weights, same logged losses,
    bit for bit -- with pageable and with pinned host batches, and with a ragged number of batches."""
    import types
    from resdepth_amd import UNet, FusedAdam, Trainer, SyntheticDsmOrthoDataset, DevicePrefetcher
    kw = dict(n_input_channels=3, start_kernel=8, depth=3, bias_conv_layer=True)
    out = {}
    for pin in (False, True):
        for pf in (0, 1, 2):
            ds = SyntheticDsmOrthoDataset(20, 3, 64, seed=3)
            train = DataLoader(ds, batch_size=4, shuffle=False, pin_memory=pin)
            torch.manual_seed(0)
            model = UNet(**kw)
            opt = FusedAdam(model.parameters(), lr=1e-3, weight_decay=1e-5)
            a = _args(tmp_path / f"p{int(pin)}{pf}", model, opt, None, train, None, n_epochs=1)
            a.prefetch_batches = pf
            tr = Trainer(a)
            m0 = tr.inference_one_epoch(0, "train")
            m1 = tr.inference_one_epoch(1, "train")
            torch.cuda.synchronize()
            out[(pin, pf)] = (m0["MAE_metric"].avg, m1["MAE_metric"].avg, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
    ref = out[(False, 0)]
    for key, (a0, a1, sd) in out.items():
        assert a0 == ref[0] and a1 == ref[1], key
        for k, v in sd.items():
            assert torch.equal(v, ref[2][k]), (key, k)
    # the iterator itself: device tensors for the five staged fields, everything else handed on, order kept
    ds = SyntheticDsmOrthoDataset(6, 3, 64, seed=4)
    plain = list(DataLoader(ds, batch_size=4, shuffle=False))
    staged = list(DevicePrefetcher(DataLoader(ds, batch_size=4, shuffle=False), "cuda:0", depth=1))
    assert len(staged) == len(plain) == 2
    for b, s_ in zip(plain, staged):
        for k in ("input", "target", "loss_mask"):
            assert s_[k].is_cuda and torch.equal(s_[k].cpu(), b[k])
        assert s_["dsm_mean"].is_cuda and s_["dsm_mean"].dtype == torch.float32
        assert not s_["patch_offset_x"].is_cuda
