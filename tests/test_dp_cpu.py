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
