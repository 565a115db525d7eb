"""resdepth_amd.plan.PlannedTrainStep (-m gpu): the training iteration of lib/Trainer.py:212-222 replayed from a recorded launch
plan (include/resdepth_hip.h: rd_plan_*) on the REAL two streams must leave exactly the bits the eager iteration leaves --
weights, BatchNorm buffers, Adam moments and step counts, losses -- across learning-rate changes and ragged batches, at full
size with the split-K ticket kernels, under data parallelism (world 2: the gradient buckets and the loss normaliser are issued by
the host between the plan's segments), and through the Trainer's loop; it enqueues in a fraction of the eager iteration's host
time; and what a plan cannot hold makes the step fall back to the eager iteration with a reason."""
import copy
import os
import sys
import time

import pytest
import torch
from torch.utils.data import DataLoader

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from test_graph_gpu import _args, _batches, _fresh, _same_state, KW, DEV  # noqa: E402

pytestmark = pytest.mark.gpu


def test_planned_iterations_leave_the_bits_of_eager_ones():
    from resdepth_amd import UNet
    from resdepth_amd.plan import PlannedTrainStep
    torch.manual_seed(0)
    sd0 = copy.deepcopy(UNet(**KW).state_dict())
    full, ragged = _batches(4, 4), _batches(3, 1, seed=40)
    seq = [full[k % 4] for k in range(12)]
    seq[7] = ragged[0]
    res = []
    for planned in (False, True):
        model, opt = _fresh(sd0)
        step = PlannedTrainStep(model, opt, warmup=2 if planned else 1 << 60)
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
    assert getattr(step, "plan_rejected", None) is None, step.plan_rejected
    assert how[:3] == ["warm-up", "warm-up", "capture preparation"] and how[3] is None
    assert how[7] == "batch shape differs from the captured one" and how[8] is None and step.replays == 8
    assert step.n_launches > 40 and step.n_segments == 1
    assert torch.equal(le, lg), (le, lg)
    _same_state(me, oe, mg, og)
    assert float(og.state_dict()["state"][0]["step"]) == 12.0


def test_full_size_step_is_bit_identical_and_cheap_to_enqueue():
    """cfg-S at batch 2 (two-stream backward, composed tail, split-K bottleneck kernels with their ticket counters, split2h
    magnitude slots zeroed by rd_zero inside the plan)."""
    from resdepth_amd import UNet
    from resdepth_amd.plan import PlannedTrainStep
    kw = dict(n_input_channels=3, start_kernel=64, depth=5, bias_conv_layer=True)
    torch.manual_seed(0)
    sd0 = copy.deepcopy(UNet(**kw).state_dict())
    bs = _batches(2, 2, t=256)
    res = []
    for planned in (False, True):
        model, opt = _fresh(sd0, kw, lr=2e-4, weight_decay=1e-5)
        step = PlannedTrainStep(model, opt, warmup=1 if planned else 1 << 60)
        losses = [step(*bs[k % 2]).clone() for k in range(7)]
        torch.cuda.synchronize()
        res.append((model, opt, torch.stack(losses).cpu(), step))
    (me, oe, le, _), (mg, og, lg, step) = res
    assert getattr(step, "plan_rejected", None) is None, step.plan_rejected
    assert step.replays == 5 and torch.equal(le, lg)
    _same_state(me, oe, mg, og)
    host = []
    for k in range(10):
        torch.cuda.synchronize()                 # an empty queue: the call's own enqueue time, not the GPU's back-pressure
        t0 = time.perf_counter()
        step(*bs[k % 2])
        host.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    # ~110 hipLaunchKernel calls from C + Adam's scalar block: measured 0.6 ms (the eager iteration: 3.5-4.3 ms)
    assert step.replays == 15 and sorted(host)[len(host) // 2] < 1.5e-3, host


def test_what_a_plan_cannot_hold_runs_eagerly_with_a_reason():
    from resdepth_amd import UNet, FusedAdam, FusedSGD
    from resdepth_amd.plan import PlannedTrainStep
    torch.manual_seed(0)
    b = _batches(2, 1)[0]
    model = UNet(**KW).to(DEV).train()
    step = PlannedTrainStep(model, FusedSGD(model.parameters(), lr=1e-3), warmup=1)
    for _ in range(3):
        step(*b)
    assert step.replays == 0 and "FusedSGD" in step.why_eager
    for kw, word in ((dict(KW, act_fn_decoder="prelu"), "PReLU"), (dict(KW, up_mode="bilinear"), "bilinear"),
                     (dict(KW, outer_skip_BN=True), "outer-skip"), (dict(KW, start_kernel=6), "twin")):
        torch.manual_seed(0)
        m = UNet(**kw).to(DEV).train()
        step = PlannedTrainStep(m, FusedAdam(m.parameters(), lr=1e-3), warmup=1)
        ref_m = copy.deepcopy(m)
        ref_o = FusedAdam(ref_m.parameters(), lr=1e-3)
        from resdepth_amd import masked_l1_loss
        for _ in range(3):
            step(*b)
            loss = masked_l1_loss(ref_m(b[0]), *b[1:])
            loss.backward()
            ref_o.step()
            for p in ref_m.parameters():
                p.grad = None
        assert step.replays == 0 and word in step.why_eager, (word, step.why_eager)
        for (k, v), (_, w) in zip(m.state_dict().items(), ref_m.state_dict().items()):
            assert torch.equal(v, w), k


@pytest.mark.parametrize("bucket_mb", [16, 2])
def test_world2_planned_iterations_equal_eager_ones_under_data_parallelism(tmp_path, bucket_mb):
    """Two ranks on the one GPU (gloo, host-staged collectives with RCCL's stream semantics): the loss normaliser's all-reduce and
    every gradient bucket's all-reduce are host actions between the plan's segments -- same losses and final state, bit for
    bit, as the eager data-parallel iteration."""
    from test_dp_world2_gpu import run_world
    outs = {}
    for plan in (0, 1):
        d = tmp_path / f"p{plan}"
        d.mkdir()
        outs[plan] = run_world(d, "plan", coll="staged", batch=4, steps=7, bucket_mb=bucket_mb, plan=plan, timeout=900)
    for r in range(2):
        e, p = outs[0][r], outs[1][r]
        assert p["rejected"] is None, p["rejected"]
        assert e["replays"] == 0 and p["replays"] == 5, (p["how"], p["replays"])
        assert p["segments"] >= 2 + p["n_buckets"], (p["segments"], p["n_buckets"])     # loss normaliser, buckets, final wait
        assert torch.equal(e["losses"], p["losses"]), (r, e["losses"], p["losses"])
        assert e["opt_step"] == p["opt_step"] == 7.0
        for k, v in e["state"].items():
            assert torch.equal(v, p["state"][k]), (r, k)
    for k, v in outs[1][0]["state"].items():            # and both ranks hold the same model
        if v.dtype.is_floating_point and "running" not in k:
            assert torch.equal(v, outs[1][1]["state"][k]), k


@pytest.mark.parametrize("prefetch", [0, 1])
def test_trainer_loop_with_a_launch_plan_equals_the_eager_loop(tmp_path, prefetch):
    from resdepth_amd import UNet, FusedAdam, Trainer, SyntheticDsmOrthoDataset
    kw = dict(n_input_channels=2, start_kernel=8, depth=2, bias_conv_layer=True)
    torch.manual_seed(0)
    sd0 = copy.deepcopy(UNet(**kw).state_dict())
    res = []
    for planned in (False, True):
        ds = SyntheticDsmOrthoDataset(22, 2, 32, seed=3)
        train = DataLoader(ds, batch_size=4, shuffle=False)                 # 5 full batches + one of 2
        val = DataLoader(SyntheticDsmOrthoDataset(6, 2, 32, seed=4), batch_size=4, shuffle=False)
        model = UNet(**kw)
        model.load_state_dict(sd0)
        opt = FusedAdam(model.parameters(), lr=1e-3, weight_decay=1e-5)
        tr = Trainer(_args(tmp_path / ("p" if planned else "e"), model, opt, train, val, 2, launch_plan=planned, prefetch_batches=prefetch))
        tr.train()
        last = torch.load(os.path.join(str(tmp_path / ("p" if planned else "e")), "checkpoints", "Model_last.pth"), weights_only=False)
        res.append((tr, last))
    (te, le), (tg, lg) = res
    assert tg._graphed is not None and tg._graphed.replays >= 6 and getattr(tg._graphed, "plan_rejected", None) is None
    assert le["loss_train"] == lg["loss_train"] and le["loss_val"] == lg["loss_val"]
    for k, v in le["model_state_dict"].items():
        assert torch.equal(v.cpu(), lg["model_state_dict"][k].cpu()), k
    _same_state(te.model, te.optimizer, tg.model, tg.optimizer)
