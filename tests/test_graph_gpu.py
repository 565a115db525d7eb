"""resdepth_amd.graph.GraphedTrainStep (-m gpu): the training iteration of lib/Trainer.py:212-222 replayed as one captured
HIP graph must leave exactly the bits the eager iteration leaves -- weights, BatchNorm buffers, Adam moments and step counts,
losses -- across learning-rate changes, ragged batches, checkpoint round trips and the Trainer's own loop."""
import copy
import os
import types

import pytest
import torch
from torch.utils.data import DataLoader

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
KW = dict(n_input_channels=3, start_kernel=16, depth=3, bias_conv_layer=True)


def _batches(n, count, t=64, c=3, seed=11):
    from resdepth_amd import synthetic_batch
    out = []
    for i in range(count):
        b = synthetic_batch(n, c, t, seed=seed + i)
        out.append((b["input"].to(DEV), b["target"].to(DEV), b["loss_mask"].to(DEV), b["dsm_mean"].to(torch.float32).to(DEV),
                    b["dsm_std"].to(torch.float32).to(DEV)))
    return out


def _fresh(sd0, kw=KW, **opt_kw):
    from resdepth_amd import UNet, FusedAdam
    model = UNet(**kw)
    model.load_state_dict(sd0)
    model = model.to(DEV).train()
    return model, FusedAdam(model.parameters(), **(opt_kw or dict(lr=2e-3, weight_decay=1e-5)))


def _same_state(ma, oa, mb, ob):
    for k, v in ma.state_dict().items():
        assert torch.equal(v, mb.state_dict()[k]), k
    sa, sb = oa.state_dict(), ob.state_dict()
    assert sa["param_groups"] == sb["param_groups"]
    for i in sa["state"]:
        for k in sa["state"][i]:
            assert torch.equal(sa["state"][i][k].cpu(), sb["state"][i][k].cpu()), (i, k)


def test_replayed_iterations_leave_the_bits_of_eager_ones():
    """12 iterations over 4 batches, the learning rate halved in the middle (what a scheduler does), one ragged batch on the way
    (eager fallback between replays): weights, BN buffers, moments, step counts and every loss bit-identical."""
    from resdepth_amd import UNet, GraphedTrainStep, _lib
    # a process that has used many streams already: the split-K scratch registry is full, a new stream evicts the least recently
    # used registration WITH a device synchronisation -- the capture stream must not go through that (it owns a pinned scratch)
    streams = [torch.cuda.Stream() for _ in range(_lib.SPLITK_MAX_STREAMS + 1)]
    for st in streams:
        with torch.cuda.stream(st):
            _lib.ensure_splitk_workspace(DEV)
    torch.manual_seed(0)
    sd0 = copy.deepcopy(UNet(**KW).state_dict())
    full, ragged = _batches(4, 4), _batches(3, 1, seed=40)
    seq = [full[k % 4] for k in range(12)]
    seq[7] = ragged[0]
    res = []
    for graphed in (False, True):
        model, opt = _fresh(sd0)
        step = GraphedTrainStep(model, opt, warmup=2 if graphed else 1 << 60)
        losses, how = [], []
        for k, b in enumerate(seq):
            if k == 6:
                opt.param_groups[0]["lr"] *= 0.5
            losses.append(step(*b).clone())
            how.append(step.why_eager)
            assert all(p.grad is None for p in model.parameters())          # lib/Trainer.py:221-222
        torch.cuda.synchronize()
        res.append((model, opt, torch.stack(losses).cpu(), how, step))
    (me, oe, le, _, _), (mg, og, lg, how, step) = res
    assert how[:3] == ["warm-up", "warm-up", "capture preparation"] and how[3] is None
    assert how[7] == "batch shape differs from the captured one" and how[8] is None and step.replays == 8
    assert torch.equal(le, lg), (le, lg)
    _same_state(me, oe, mg, og)
    assert float(og.state_dict()["state"][0]["step"]) == 12.0


def test_full_size_step_is_bit_identical_and_one_launch():
    """cfg-S at batch 2 (two-stream backward, composed tail, split-K bottleneck kernels with their ticket counters): replays leave
    the same bits as eager iterations, and a replayed step enqueues in well under a millisecond."""
    import time
    from resdepth_amd import UNet, GraphedTrainStep
    kw = dict(n_input_channels=3, start_kernel=64, depth=5, bias_conv_layer=True)
    torch.manual_seed(0)
    sd0 = copy.deepcopy(UNet(**kw).state_dict())
    bs = _batches(2, 2, t=256)
    res = []
    for graphed in (False, True):
        model, opt = _fresh(sd0, kw, lr=2e-4, weight_decay=1e-5)
        step = GraphedTrainStep(model, opt, warmup=1 if graphed else 1 << 60)
        losses = [step(*bs[k % 2]).clone() for k in range(7)]
        torch.cuda.synchronize()
        res.append((model, opt, torch.stack(losses).cpu(), step))
    (me, oe, le, _), (mg, og, lg, step) = res
    assert step.replays == 5 and torch.equal(le, lg)
    _same_state(me, oe, mg, og)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(10):
        step(*bs[k % 2])
    host = (time.perf_counter() - t0) / 10
    torch.cuda.synchronize()
    assert step.replays == 15 and host < 2e-3, host           # measured 0.35 ms (the eager iteration: 3.5 ms)


def test_checkpoint_round_trip_and_load_state_dict_recapture():
    """optimizer.state_dict() taken between replays carries the right step counts; load_state_dict() into the graphed optimizer
    drops the capture (its moment buffers are replaced) and the next calls capture again -- same bits as an eager run that
    loads the same checkpoint."""
    from resdepth_amd import UNet, GraphedTrainStep
    torch.manual_seed(0)
    sd0 = copy.deepcopy(UNet(**KW).state_dict())
    bs = _batches(4, 3)
    ckpt = {}
    res = []
    for graphed in (False, True):
        model, opt = _fresh(sd0)
        step = GraphedTrainStep(model, opt, warmup=1 if graphed else 1 << 60)
        for k in range(5):
            step(*bs[k % 3])
        sd = copy.deepcopy(opt.state_dict())
        assert float(sd["state"][0]["step"]) == 5.0
        if not graphed:
            ckpt["opt"], ckpt["model"] = sd, copy.deepcopy(model.state_dict())
        else:
            for i in sd["state"]:
                for k in sd["state"][i]:
                    assert torch.equal(sd["state"][i][k].cpu(), ckpt["opt"]["state"][i][k].cpu()), (i, k)
        # roll both back to the eager run's checkpoint and continue
        model.load_state_dict(ckpt["model"])
        opt.load_state_dict(copy.deepcopy(ckpt["opt"]))
        before = step.replays
        losses = [step(*bs[k % 3]).clone() for k in range(5)]
        torch.cuda.synchronize()
        if graphed:
            assert step.replays > before                    # captured again after the load
        res.append((model, opt, torch.stack(losses).cpu()))
    (me, oe, le), (mg, og, lg) = res
    assert torch.equal(le, lg)
    _same_state(me, oe, mg, og)
    assert float(og.state_dict()["state"][0]["step"]) == 10.0


def test_ineligible_configurations_run_eagerly_with_a_reason():
    from resdepth_amd import UNet, FusedSGD, GraphedTrainStep
    torch.manual_seed(0)
    model = UNet(**KW).to(DEV).train()
    step = GraphedTrainStep(model, FusedSGD(model.parameters(), lr=1e-3), warmup=1)
    b = _batches(2, 1)[0]
    for _ in range(3):
        step(*b)
    assert step.replays == 0 and "FusedSGD" in step.why_eager
    model.eval()
    from resdepth_amd import FusedAdam
    step = GraphedTrainStep(model, FusedAdam(model.parameters(), lr=1e-3), warmup=1)
    # an eval-mode model has no training iteration to capture; the call itself is the caller's mistake, reported as such
    assert step._eligible(b[0]) == "model in eval mode"


def _args(tmp, model, opt, train, val, n_epochs, **extra):
    return types.SimpleNamespace(
        model=model, optimizer=opt, scheduler=torch.optim.lr_scheduler.StepLR(opt, step_size=1, gamma=0.5),
        criterion=torch.nn.L1Loss(reduction="mean"), trainloader=train, valloader=val, n_epochs=n_epochs, evaluate_rate=1,
        save_model_rate=1, freq_average_train_loss=3, save_dir=str(tmp), log_file=os.path.join(str(tmp), "training.log"),
        checkpoint_dir=os.path.join(str(tmp), "checkpoints"), tboard_log_dir=os.path.join(str(tmp), "tb"), pretrained_path=None, **extra)


@pytest.mark.parametrize("prefetch", [0, 1])
def test_trainer_loop_with_hip_graph_equals_the_eager_loop(tmp_path, prefetch):
    """Trainer(hip_graph=True): two epochs over a loader with a ragged last batch, StepLR between the epochs -- the same weights,
    moments and logged epoch losses as the eager Trainer, with and without the host -> device prefetch."""
    from resdepth_amd import UNet, FusedAdam, Trainer, SyntheticDsmOrthoDataset
    kw = dict(n_input_channels=2, start_kernel=8, depth=2, bias_conv_layer=True)
    torch.manual_seed(0)
    sd0 = copy.deepcopy(UNet(**kw).state_dict())
    res = []
    for graphed in (False, True):
        ds = SyntheticDsmOrthoDataset(22, 2, 32, seed=3)
        train = DataLoader(ds, batch_size=4, shuffle=False)                 # 5 full batches + one of 2
        val = DataLoader(SyntheticDsmOrthoDataset(6, 2, 32, seed=4), batch_size=4, shuffle=False)   # eval-mode forwards between the epochs
        model = UNet(**kw)
        model.load_state_dict(sd0)
        opt = FusedAdam(model.parameters(), lr=1e-3, weight_decay=1e-5)
        tr = Trainer(_args(tmp_path / ("g" if graphed else "e"), model, opt, train, val, 2, hip_graph=graphed, prefetch_batches=prefetch))
        tr.train()
        last = torch.load(os.path.join(str(tmp_path / ("g" if graphed else "e")), "checkpoints", "Model_last.pth"), weights_only=False)
        res.append((tr, last))
    (te, le), (tg, lg) = res
    assert tg._graphed is not None and tg._graphed.replays >= 6
    assert le["loss_train"] == lg["loss_train"] and le["loss_val"] == lg["loss_val"]
    for k, v in le["model_state_dict"].items():
        assert torch.equal(v.cpu(), lg["model_state_dict"][k].cpu()), k
    _same_state(te.model, te.optimizer, tg.model, tg.optimizer)


def test_copy_segments_moves_a_batch_in_one_launch_and_falls_back_for_what_it_cannot_take():
    """rd_copy_segments (the new batch -> the static input buffers of a captured / planned iteration, lib/Trainer.py:212-215): up to
    eight aligned ranges in one launch; odd sizes, other dtypes' layouts or a ninth pair go through torch -- same result."""
    from resdepth_amd import _lib
    g = torch.Generator(device="cuda").manual_seed(1)
    shapes = [(32, 3, 256, 256), (32, 1, 256, 256), (32, 1, 256, 256), (32,), (32,), (7, 3), (5,), (4, 4), (16, 16), (3, 5, 7)]
    src = [torch.randn(s, device="cuda", generator=g) for s in shapes]
    src[2] = src[2] > 0                                      # a bool mask: one byte per element, still whole 16-byte units
    src.append(torch.randn(40, device="cuda", generator=g)[1:33])          # misaligned view: torch's copy
    dst = [torch.empty_like(s) for s in src]
    _lib.copy_segments(list(zip(dst, src)))
    torch.cuda.synchronize()
    for d, s in zip(dst, src):
        assert torch.equal(d, s)
    with pytest.raises(RuntimeError, match="1..8 segments"):
        _lib.check(_lib.load().rd_copy_segments(None, None, None, 0, None), "copy_segments")
