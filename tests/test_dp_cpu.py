"""Data-parallel host logic on CPU: world_size-2 `gloo` processes (the no-GPU stand-in for RCCL).

The compute inside the workers is the ORACLE (tests may use it); what is under test is
resdepth_amd.dp: bucket planning, asynchronous bucketed all-reduce driven by per-parameter readiness
in backward order, the global loss normaliser, SyncBN statistic merging and batch sharding.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import unet_oracle as O
from resdepth_amd import dp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class FlatModel:
    """Minimal stand-in for resdepth_amd.UNet's flat-gradient interface."""

    def __init__(self, shapes):
        self.params = [torch.nn.Parameter(torch.zeros(s)) for s in shapes]
        self._offsets, o = [], 0
        for p in self.params:
            self._offsets.append(o)
            o += p.numel()
        self._flat_grad = torch.zeros(o)

    def parameters(self):
        return iter(self.params)


def test_plan_buckets_tiles_the_buffer_from_the_end():
    sizes = [10, 3, 50, 7, 30, 1]
    offs = [0, 10, 13, 63, 70, 100]
    b = dp.GradSync.plan_buckets(offs, sizes, 32)
    assert b[0]["hi"] == 101 and b[-1]["lo"] == 0
    for x, y in zip(b[:-1], b[1:]):
        assert x["lo"] == y["hi"]
    assert set().union(*[x["params"] for x in b]) == set(range(6))
    assert all(x["hi"] - x["lo"] >= 32 for x in b[:-1])
    # tail cap: the last bucket [0, 13) = tensors 0 (10) and 1 (3) is cut so that its final piece -- the one whose all-reduce
    # nothing overlaps -- holds at most 10 elements: [tensor 1] goes out earlier, [tensor 0] stays the tail
    c = dp.GradSync.plan_buckets(offs, sizes, 32, tail_elems=10)
    assert [(x["lo"], x["hi"]) for x in c[:-2]] == [(x["lo"], x["hi"]) for x in b[:-1]]
    assert (c[-2]["lo"], c[-2]["hi"], c[-2]["params"]) == (10, 13, {1}) and (c[-1]["lo"], c[-1]["hi"], c[-1]["params"]) == (0, 10, {0})
    for x, y in zip(c[:-1], c[1:]):
        assert x["lo"] == y["hi"]
    # a tail that already fits, or that is one tensor, is left alone; a cap smaller than the last tensor keeps that tensor whole
    assert dp.GradSync.plan_buckets(offs, sizes, 32, tail_elems=13) == b
    assert dp.GradSync.plan_buckets(offs, sizes, 32, tail_elems=4)[-1]["params"] == {0}
    # geometric: a long leftover is cut repeatedly, caps 2, 8, 32 from the start of the buffer
    sizes2, offs2 = [2, 6, 20, 40, 100], [0, 2, 8, 28, 68]
    g = dp.GradSync.plan_buckets(offs2, sizes2, 90, tail_elems=2)
    assert [(x["lo"], x["hi"]) for x in g] == [(68, 168), (28, 68), (8, 28), (2, 8), (0, 2)]
    assert [x["params"] for x in g] == [{4}, {3}, {2}, {1}, {0}]


def test_shard_batch():
    b = O.synthetic_batch(4, 2, 8)
    b["tag"] = "x"
    s1 = dp.shard_batch(b, 1, 2)
    assert torch.equal(s1["input"], b["input"][2:4]) and torch.equal(s1["dsm_std"], b["dsm_std"][2:4]) and s1["tag"] == "x"
    with pytest.raises(ValueError):
        dp.shard_batch(b, 0, 3)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")        # the container hostname may not resolve
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        spec = O.Spec(n_input_channels=2, start_kernel=4, depth=2, bias_conv_layer=True)
        sd = O.init_state_dict(spec, 0)
        for k in sd:                       # non-trivial running stats so eval-mode BN does something
            if k.endswith("running_mean"):
                sd[k] = torch.linspace(-0.2, 0.2, sd[k].numel())
            if k.endswith("running_var"):
                sd[k] = torch.linspace(0.5, 1.5, sd[k].numel())
        keys = O.param_keys(spec)
        full = O.synthetic_batch(4, 2, 16, seed=5)
        full["dsm_std"] = torch.tensor([1.0, 2.0, 3.0, 0.5])

        def grads_of(batch, count=None):
            leaves = {k: sd[k].clone().requires_grad_(True) for k in keys}
            work = dict(sd)
            work.update(leaves)
            yp = O.forward(work, batch["input"], spec, training=False)
            s = batch["dsm_std"].view(-1, 1, 1, 1)
            num = (((yp - batch["target"]) * s).abs() * batch["loss_mask"]).sum()
            cnt = batch["loss_mask"].sum() if count is None else count
            return num.detach().double(), torch.autograd.grad(num / cnt, [leaves[k] for k in keys])

        _, ref = grads_of(full)                                   # single-device, whole batch
        gs = dp.GradSync(bucket_bytes=256)
        local = dp.shard_batch(full, rank, world)
        num_local, _ = grads_of(local)
        sums = torch.stack([num_local, local["loss_mask"].sum().double()])
        numel = gs.allreduce_loss_sums(sums, local["target"].numel())
        assert numel == full["target"].numel()
        assert float(sums[1]) == float(full["loss_mask"].sum())
        _, gl = grads_of(local, count=sums[1].float())           # local gradient with the GLOBAL normaliser
        model = FlatModel([tuple(sd[k].shape) for k in keys])
        for i, g in enumerate(gl):
            o = model._offsets[i]
            model._flat_grad[o:o + g.numel()] = g.flatten()
        for i in reversed(range(len(keys))):                      # backward order: last parameter first
            gs.params_ready(model, [i])
        assert sum(gs._launched) >= len(gs._buckets) - 1          # buckets were issued before finish()
        gs.finish(model)
        for i, g in enumerate(ref):
            o = model._offsets[i]
            got = model._flat_grad[o:o + g.numel()].view(g.shape)
            err = float((got - g).norm() / (g.norm() + 1e-30))
            assert err < 2e-6, (keys[i], err)
        # SyncBN statistic merge: per-rank (sum, sum^2) all-reduced == whole-batch statistics
        z = torch.randn(4, 6, 8, 8, generator=torch.Generator().manual_seed(1)) * 2 + 1
        zl = z[rank * 2:(rank + 1) * 2]
        st = torch.cat([zl.double().sum((0, 2, 3)), (zl.double() ** 2).sum((0, 2, 3))])
        cnt = gs.allreduce_stats(st, zl.numel() // 6)
        mean = st[:6] / cnt
        var = st[6:] / cnt - mean ** 2
        assert torch.allclose(mean, z.double().mean((0, 2, 3)), atol=1e-12)
        assert torch.allclose(var, z.double().var((0, 2, 3), unbiased=False), atol=1e-10)
        # broadcast_parameters: rank 1 adopts rank 0's values
        lin = torch.nn.Linear(3, 2)
        with torch.no_grad():
            lin.weight.fill_(float(rank + 1))
        dp.broadcast_parameters(lin, src=0)
        assert float(lin.weight[0, 0]) == 1.0
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "fail: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_gradient_sync_matches_single_device():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in results), results


class _SyncBN(torch.autograd.Function):
    """The engine's host-side SyncBN protocol (resdepth_amd/unet.py: _bn_forward / bn_backward with sync_bn) on CPU
    tensors: forward exchanges (sum z, sum z^2) through GradSync.allreduce_stats, backward exchanges (sum g, sum g*xhat)
    through GradSync.allreduce_sums; the affine-parameter gradients stay LOCAL (the gradient all-reduce sums them)."""

    @staticmethod
    def forward(ctx, z, gamma, beta, gs):
        c = z.shape[1]
        sums = torch.cat([z.double().sum((0, 2, 3)), (z.double() ** 2).sum((0, 2, 3))])
        count = gs.allreduce_stats(sums, z.numel() // c)
        mean = (sums[:c] / count).float()
        var = (sums[c:] / count - (sums[:c] / count) ** 2).clamp_min(0).float()
        invstd = torch.rsqrt(var + 1e-5)
        xhat = (z - mean.view(1, c, 1, 1)) * invstd.view(1, c, 1, 1)
        ctx.save_for_backward(xhat, gamma, invstd)
        ctx.gs, ctx.count = gs, count
        return xhat * gamma.view(1, c, 1, 1) + beta.view(1, c, 1, 1)

    @staticmethod
    def backward(ctx, g):
        xhat, gamma, invstd = ctx.saved_tensors
        c = g.shape[1]
        sums = torch.cat([g.double().sum((0, 2, 3)), (g.double() * xhat.double()).sum((0, 2, 3))])
        dbeta, dgamma = sums[:c].float().clone(), sums[c:].float().clone()          # local sums = parameter gradients
        ctx.gs.allreduce_sums(sums)
        k1 = (sums[:c] / ctx.count).float().view(1, c, 1, 1)
        k2 = (sums[c:] / ctx.count).float().view(1, c, 1, 1)
        dz = (gamma * invstd).view(1, c, 1, 1) * (g - k1 - xhat * k2)
        return dz, dgamma, dbeta, None


def _worker4(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")        # the container hostname may not resolve
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        import torch.nn.functional as F
        g = torch.Generator().manual_seed(11)
        w1 = torch.randn(6, 2, 3, 3, generator=g) * 0.3
        gamma, beta = torch.rand(6, generator=g) + 0.5, torch.randn(6, generator=g) * 0.1
        w2 = torch.randn(1, 6, 3, 3, generator=g) * 0.3
        b2 = torch.randn(1, generator=g)
        full = O.synthetic_batch(8, 2, 16, seed=9)
        full["dsm_std"] = torch.linspace(0.5, 3.0, 8)

        def net(x, p, bn):
            z = F.conv2d(x, p[0], None, 1, 1)
            a = torch.relu(bn(z, p[1], p[2]))
            return x[:, 0:1] + F.conv2d(a, p[3], p[4], 1, 1)

        def loss_num(yp, b):
            return (((yp - b["target"]) * b["dsm_std"].view(-1, 1, 1, 1)).abs() * b["loss_mask"]).sum()

        # reference: one process, whole batch, torch's own training-mode batch norm
        pr = [t.clone().requires_grad_(True) for t in (w1, gamma, beta, w2, b2)]
        ref_loss = loss_num(net(full["input"], pr, lambda z, ga, be: F.batch_norm(z, None, None, ga, be, True, 0.1, 1e-5)), full) / \
            full["loss_mask"].sum()
        ref = torch.autograd.grad(ref_loss, pr)
        # 4 ranks, 2 tiles each: SyncBN + global loss normaliser + bucketed gradient all-reduce
        gs = dp.GradSync(bucket_bytes=64)                          # 16-float buckets: five parameters -> several buckets
        local = dp.shard_batch(full, rank, world)
        pl = [t.clone().requires_grad_(True) for t in (w1, gamma, beta, w2, b2)]
        yp = net(local["input"], pl, lambda z, ga, be: _SyncBN.apply(z, ga, be, gs))
        num = loss_num(yp, local)
        sums = torch.stack([num.detach().double(), local["loss_mask"].sum().double()])
        gs.allreduce_loss_sums(sums, local["target"].numel())
        assert float(sums[1]) == float(full["loss_mask"].sum())
        assert abs(float(sums[0] / sums[1]) - float(ref_loss)) <= 1e-5 * abs(float(ref_loss))
        gl = torch.autograd.grad(num / sums[1].float(), pl)
        model = FlatModel([tuple(t.shape) for t in pl])
        for i, gr in enumerate(gl):
            o = model._offsets[i]
            model._flat_grad[o:o + gr.numel()] = gr.flatten()
        # readiness arrives OUT of backward order (uneven completion): buckets fire as soon as all their members are in,
        # in the same order on every rank because every rank makes the same calls
        order = [2, 4, 0, 3, 1]
        launched_before = []
        for i in order:
            gs.params_ready(model, [i])
            launched_before.append(sum(gs._launched))
        assert launched_before[-1] >= 1 and launched_before == sorted(launched_before)
        gs.finish(model)
        for i, gr in enumerate(ref):
            o = model._offsets[i]
            got = model._flat_grad[o:o + gr.numel()].view(gr.shape)
            err = float((got - gr).norm() / (gr.norm() + 1e-30))
            assert err < 2e-5, (i, err)
        # a second step re-uses the plan (buckets reset by finish)
        for i in range(5):
            gs.params_ready(model, [4 - i])
        gs.finish(model)
        # unequal shards are refused on every rank instead of deadlocking later
        gs.check_equal_across_ranks(7, "len(loader)")
        try:
            gs.check_equal_across_ranks(7 + (rank == 2), "len(loader)")
            raise AssertionError("unequal loader lengths were accepted")
        except RuntimeError as e:
            assert "differs across ranks" in str(e)
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "fail: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_four_rank_gloo_syncbn_uneven_bucket_order_matches_single_device():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker4, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in results), results


def _worker8(rank, world, port, q, sync_bn):
    """world 8, one tile per rank: the protocol of the 8-GPU node (BASELINE configs[2]) executed end to end on CPU -- global loss
    normaliser, bucketed gradient all-reduce with the geometric tail (three pieces), SyncBN on (== one process at the global
    batch) or off (== the sum of the eight shards' local-BN gradients under the global normaliser)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        import torch.nn.functional as F
        g = torch.Generator().manual_seed(21)
        w1 = torch.randn(8, 2, 3, 3, generator=g) * 0.3
        gamma, beta = torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g) * 0.1
        w2 = torch.randn(8, 8, 3, 3, generator=g) * 0.2
        w3 = torch.randn(1, 8, 3, 3, generator=g) * 0.3
        b3 = torch.randn(1, generator=g)
        base = (w1, gamma, beta, w2, w3, b3)
        full = O.synthetic_batch(world, 2, 16, seed=19)
        full["dsm_std"] = torch.linspace(0.5, 3.0, world)

        def net(x, p, bn):
            a = torch.relu(bn(F.conv2d(x, p[0], None, 1, 1), p[1], p[2]))
            a = torch.relu(F.conv2d(a, p[3], None, 1, 1))
            return x[:, 0:1] + F.conv2d(a, p[4], p[5], 1, 1)

        def loss_num(yp, b):
            return (((yp - b["target"]) * b["dsm_std"].view(-1, 1, 1, 1)).abs() * b["loss_mask"]).sum()

        local_bn = lambda z, ga, be: F.batch_norm(z, None, None, ga, be, True, 0.1, 1e-5)      # noqa: E731
        cnt_all = full["loss_mask"].sum()
        if sync_bn:          # reference: one process, the whole batch
            pr = [t.clone().requires_grad_(True) for t in base]
            ref = torch.autograd.grad(loss_num(net(full["input"], pr, local_bn), full) / cnt_all, pr)
        else:                # reference: every shard on its own statistics, gradients summed
            ref = None
            for r in range(world):
                sh = dp.shard_batch(full, r, world)
                pr = [t.clone().requires_grad_(True) for t in base]
                # one tile per rank: 256 values per channel, batch_norm is happy
                gr = torch.autograd.grad(loss_num(net(sh["input"], pr, local_bn), sh) / cnt_all, pr)
                ref = [x.double() for x in gr] if ref is None else [a + x.double() for a, x in zip(ref, gr)]
        # 64-byte buckets, 32-byte geometric tail: the leftover bucket is cut again (tests/test_dp_cpu.py::test_plan_buckets...)
        gs = dp.GradSync(bucket_bytes=2048, tail_bytes=64)
        local = dp.shard_batch(full, rank, world)
        pl = [t.clone().requires_grad_(True) for t in base]
        bn = (lambda z, ga, be: _SyncBN.apply(z, ga, be, gs)) if sync_bn else local_bn
        num = loss_num(net(local["input"], pl, bn), local)
        sums = torch.stack([num.detach().double(), local["loss_mask"].sum().double()])
        assert gs.allreduce_loss_sums(sums, local["target"].numel()) == full["target"].numel()
        assert float(sums[1]) == float(cnt_all)
        gl = torch.autograd.grad(num / sums[1].float(), pl)
        model = FlatModel([tuple(t.shape) for t in pl])
        for i, gr in enumerate(gl):
            o = model._offsets[i]
            model._flat_grad[o:o + gr.numel()] = gr.flatten()
        for i in reversed(range(len(pl))):
            gs.params_ready(model, [i])
        n_b = len(gs._buckets)
        assert n_b >= 3 and sum(gs._launched) >= n_b - 1, (n_b, gs._launched)
        sizes = [b["hi"] - b["lo"] for b in gs._buckets]
        # [b3, w3, w2] fill the first bucket; the leftover [w1, gamma, beta] is cut again: [gamma, beta] goes out before the
        # last (whole-tensor) piece w1
        assert sizes == [649, 16, 144] and sum(sizes) == sum(t.numel() for t in pl), sizes
        gs.finish(model)
        for i, gr in enumerate(ref):
            o = model._offsets[i]
            got = model._flat_grad[o:o + gr.numel()].view(gr.shape)
            err = float((got.double() - gr.double()).norm() / (gr.double().norm() + 1e-30))
            assert err < 3e-5, (i, err)
        gs.check_equal_across_ranks(3, "len(loader)")
        lin = torch.nn.Linear(3, 2)
        with torch.no_grad():
            lin.weight.fill_(float(rank + 1))
        dp.broadcast_parameters(lin, src=0)
        assert float(lin.weight[0, 0]) == 1.0
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "fail: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("sync_bn", [True, False])
def test_eight_rank_gloo_protocol_matches_single_device(sync_bn):
    """r05 verdict, missing item 4: a world-8 execution of the DP protocol (the collectives, not only the plans)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, q, sync_bn)) for r in range(8)]
    for p in procs:
        p.start()
    results = [q.get(timeout=400) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in results), results
