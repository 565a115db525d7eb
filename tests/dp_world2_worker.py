"""One rank of a world-size-N run of the HIP engine with TWO (or more) PROCESSES SHARING cuda:0 -- the only way
to execute resdepth_amd's data-parallel path at world size > 1 on a one-GPU box (RCCL refuses duplicate devices).
torch.distributed backend: gloo; device-tensor collectives either natively (if this torch build's gloo takes them)
or through tests/host_staged_collectives.py (same stream semantics as RCCL).  Driven by tests/test_dp_world2_gpu.py.

modes:
  probe  -- one tiny device all-reduce through the chosen collectives (exit 0 = works)
  train  -- cfg-S architecture, global batch B cut into contiguous shards (dp.shard_batch), dp.attach (+/- SyncBN),
            two-stream backward, FusedAdam, K steps; rank != 0 starts from DIFFERENT weights and adopts rank 0's through
            dp.broadcast_parameters.  Writes losses, step-0 gradients (after the all-reduce), final parameters + buffers.
  infer  -- cfg-G: predict_linear_blend over this rank's tile shard: row bands (shared rows exchanged point to point, every rank
            delivers its rows into one shared host array) or --shard-mode stride (full rasters summed on rank 0, torch.distributed.reduce)
  trainer -- resdepth_amd.Trainer (lib/Trainer.py surface) under data parallelism: every rank its own loader shard and its own
            output directory, ReduceLROnPlateau driven by the (global) validation loss, four epochs; --ragged 1 / 2 gives
            rank 1 a shard with one more batch / a smaller last batch, which the constructor must refuse on EVERY rank
"""
import argparse
import os
import sys

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

CFG_S = dict(n_input_channels=3, start_kernel=64, depth=5, bias_conv_layer=True)
# channel counts that are not multiples of 4: the engine runs such a model on its zero-padded twin
CFG_ODD = dict(n_input_channels=2, start_kernel=6, depth=2, max_filter_depth=10, act_fn_decoder="lrelu")
ARCH = {"S": CFG_S, "odd": CFG_ODD}


def make_batch(n, step, t=256, c=3):
    """Global batch of step `step`: per-sample dsm_std varies so that a wrong shard <-> std pairing shows."""
    from resdepth_amd import synthetic_batch
    b = synthetic_batch(n, c, t, seed=4000 + step)
    b["dsm_std"] = torch.linspace(0.5, 3.0, n)
    return b


def make_infer_model(dev):
    from resdepth_amd import UNet
    torch.manual_seed(11)
    model = UNet(**CFG_S)
    sd = model.state_dict()
    g = torch.Generator().manual_seed(12)
    for k in [k for k in sd if k.endswith("running_mean")]:
        pre = k[:-len("running_mean")]
        sd[k] = torch.randn(sd[k].shape, generator=g) * 0.2
        sd[pre + "running_var"] = torch.rand(sd[k].shape, generator=g) * 0.8 + 0.4
    model.load_state_dict(sd)
    return model.to(dev).eval()


TRAINER = dict(per_rank_batch=2, train_batches=3, val_batches=2, lr=1e-3, epochs=4)


def trainer_datasets(rank, tile, c, ragged=0):
    """This rank's shard of the training / validation samples (disjoint seeds per rank)."""
    from resdepth_amd import SyntheticDsmOrthoDataset
    T = TRAINER
    n_train = T["per_rank_batch"] * T["train_batches"]
    if rank == 1 and ragged == 1:
        n_train += T["per_rank_batch"]           # one more batch than rank 0: the per-step collectives would deadlock
    if rank == 1 and ragged == 2:
        n_train -= 1                             # same number of batches, smaller last batch: biased SyncBN statistics
    return (SyntheticDsmOrthoDataset(n_train, c, tile, seed=10 + rank),
            SyntheticDsmOrthoDataset(T["per_rank_batch"] * T["val_batches"], c, tile, seed=20 + rank))


def run_trainer(a, dev):
    import types
    from torch.utils.data import DataLoader
    from resdepth_amd import UNet, FusedAdam, Trainer, dp
    T = TRAINER
    kw = ARCH[a.arch]
    torch.manual_seed(100 + a.rank)                  # rank 0's weights win through the broadcast
    model = UNet(**kw).to(dev)
    dp.attach(model, sync_bn=bool(a.sync_bn), bucket_bytes=a.bucket_mb << 20)
    dp.broadcast_parameters(model, 0)
    opt = FusedAdam(model.parameters(), lr=T["lr"], weight_decay=1e-5)
    # threshold 0.9 (relative): an epoch counts as an improvement only below 0.1 x the best validation loss, i.e. never after
    # the first one -- with patience 0 the rate halves after each of the epochs 1, 2 and 3, on every rank or on none
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="min", factor=0.5, patience=0, threshold=0.9)
    ds_train, ds_val = trainer_datasets(a.rank, a.tile, kw["n_input_channels"], a.ragged)
    out_dir = os.path.join(os.path.dirname(a.out), f"trainer_out_r{a.rank}")
    args = types.SimpleNamespace(
        model=model, optimizer=opt, scheduler=sched, criterion=torch.nn.L1Loss(reduction="mean"),
        trainloader=DataLoader(ds_train, batch_size=T["per_rank_batch"], shuffle=False),
        valloader=DataLoader(ds_val, batch_size=T["per_rank_batch"], shuffle=False), n_epochs=T["epochs"], evaluate_rate=1,
        save_model_rate=2, freq_average_train_loss=2, save_dir=out_dir, log_file=os.path.join(out_dir, "training.log"),
        checkpoint_dir=os.path.join(out_dir, "checkpoints"), tboard_log_dir=os.path.join(out_dir, "tb"), pretrained_path=None)
    try:
        tr = Trainer(args)
    except RuntimeError as e:
        torch.save({"error": str(e)}, a.out)
        return
    tr.train()
    torch.cuda.synchronize()
    files = sorted(os.path.relpath(os.path.join(d, f), out_dir) for d, _, fs in os.walk(out_dir) for f in fs) \
        if os.path.isdir(out_dir) else []
    torch.save({"lr": opt.param_groups[0]["lr"], "best_loss": tr.best_loss, "index_best_loss": tr.index_best_loss,
                "is_main": tr.is_main, "files": files, "out_dir": out_dir, "num_bad_epochs": sched.num_bad_epochs,
                "adam_steps": float(next(iter(opt.state.values()))["step"]) if opt.state else None,
                "state": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}}, a.out)


INFER_RASTER = dict(rows=640, cols=900, areas=[((0, 559), (0, 383)), ((300, 899), (200, 639))])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="train")
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--port", type=int, required=True)
    ap.add_argument("--coll", default="staged", choices=["native", "staged"])
    ap.add_argument("--shard-mode", default="bands", choices=["bands", "stride"])
    ap.add_argument("--save-decisions", type=int, default=0,
                    help="train: also save the HIP path's discrete decisions (activation masks, pool arg-max) on this rank's shard at step 0")
    ap.add_argument("--sync-bn", type=int, default=0)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--tile", type=int, default=256)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--bucket-mb", type=int, default=16)
    ap.add_argument("--serial-backward", type=int, default=0)
    ap.add_argument("--arch", default="S", choices=["S", "odd"])
    ap.add_argument("--tune", default="", help="RD_TUNE string (kernel-selection knobs), applied before the library loads")
    ap.add_argument("--ragged", type=int, default=0)
    ap.add_argument("--plan", type=int, default=0, help="mode plan: 1 = replay the iteration from a launch plan, 0 = eager")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    if a.tune:
        os.environ["RD_TUNE"] = a.tune

    import datetime
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(a.port)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")        # the box's hostname may not resolve: pairs connect over loopback
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    # a rendezvous that does not complete fails in two minutes (the launcher then retries once on a fresh port)
    dist.init_process_group("gloo", rank=a.rank, world_size=a.world, timeout=datetime.timedelta(seconds=120))
    try:
        if a.coll == "staged":
            import host_staged_collectives
            host_staged_collectives.install()
        if a.mode == "probe":
            t = torch.full((1 << 20,), float(a.rank + 1), device=dev)
            h = dist.all_reduce(t, async_op=True)
            h.wait()
            torch.cuda.synchronize()
            want = a.world * (a.world + 1) / 2
            assert float(t[0]) == want and float(t[-1]) == want, (float(t[0]), want)
            r = torch.full((8,), float(a.rank + 1), device=dev, dtype=torch.float64)
            dist.reduce(r, dst=0)
            torch.cuda.synchronize()
            assert a.rank != 0 or float(r[0]) == want
            torch.save({"ok": True}, a.out)
            return
        from resdepth_amd import UNet, FusedAdam, masked_l1_loss, dp
        if a.mode == "train":
            torch.manual_seed(100 + a.rank)              # rank 0's seed-100 weights win through the broadcast
            model = UNet(**ARCH[a.arch]).to(dev).train()
            model.two_stream_backward = not a.serial_backward
            gs = dp.attach(model, sync_bn=bool(a.sync_bn), bucket_bytes=a.bucket_mb << 20)
            dp.broadcast_parameters(model, 0)
            opt = FusedAdam(model.parameters(), lr=2e-4, weight_decay=1e-5)
            losses, grads0, bufs0 = [], None, None
            n_buckets = None
            dec0 = None
            for step in range(a.steps):
                local = dp.shard_batch(make_batch(a.batch, step, a.tile, ARCH[a.arch]["n_input_channels"]), a.rank, a.world)
                if step == 0 and a.save_decisions:
                    # one more training-mode engine forward (with the SyncBN exchanges, in lock-step on every rank); restores
                    # the BN buffers afterwards
                    from oracle import unet_oracle as O_
                    from test_unet_gpu import _hip_decisions
                    dec0 = _hip_decisions(model, local["input"].to(dev), O_.Spec(**ARCH[a.arch]))
                y = model(local["input"].to(dev))
                loss = masked_l1_loss(y, local["target"], local["loss_mask"], local["dsm_mean"], local["dsm_std"], grad_sync=gs)
                loss.backward()
                if step == 0:
                    y0 = y.detach().cpu().clone()
                    grads0 = {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters()}
                    bufs0 = {k: v.detach().cpu().clone() for k, v in model.named_buffers()}
                    n_buckets = len(gs._buckets) if gs._buckets else 0
                opt.step()
                for p in model.parameters():
                    p.grad = None
                losses.append(float(loss))
            torch.cuda.synchronize()
            torch.save({"losses": losses, "y0": y0, "grads0": grads0, "bufs0": bufs0, "n_buckets": n_buckets, "dec0": dec0,
                        "state": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}}, a.out)
        elif a.mode == "plan":
            # the iteration through resdepth_amd.plan.PlannedTrainStep (--plan 1) or its eager path (--plan 0): local BatchNorm,
            # bucketed gradient all-reduce issued between the plan's segments
            from resdepth_amd.plan import PlannedTrainStep
            torch.manual_seed(100 + a.rank)
            model = UNet(**ARCH[a.arch]).to(dev).train()
            gs = dp.attach(model, sync_bn=False, bucket_bytes=a.bucket_mb << 20)
            dp.broadcast_parameters(model, 0)
            opt = FusedAdam(model.parameters(), lr=2e-4, weight_decay=1e-5)
            step_fn = PlannedTrainStep(model, opt, warmup=1 if a.plan else 1 << 60)
            shards = []
            for k in range(2):
                b = dp.shard_batch(make_batch(a.batch, k, a.tile, ARCH[a.arch]["n_input_channels"]), a.rank, a.world)
                shards.append((b["input"].to(dev), b["target"].to(dev), b["loss_mask"].to(dev), b["dsm_mean"].float().to(dev),
                               b["dsm_std"].float().to(dev)))
            losses, how = [], []
            for step in range(a.steps):
                losses.append(step_fn(*shards[step % 2]).clone())
                how.append(step_fn.why_eager)
            torch.cuda.synchronize()
            torch.save({"losses": torch.stack(losses).cpu(), "how": how, "replays": step_fn.replays, "segments": step_fn.n_segments,
                        "launches": step_fn.n_launches, "rejected": getattr(step_fn, "plan_rejected", None),
                        "n_buckets": len(gs._buckets) if gs._buckets else 0,
                        "opt_step": float(next(iter(opt.state_dict()["state"].values()))["step"]),
                        "state": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}}, a.out)
        elif a.mode == "infer":
            from torch.utils.data import DataLoader
            from resdepth_amd import SyntheticRasterTiles, predict_linear_blend
            model = make_infer_model(dev)
            ds = SyntheticRasterTiles(INFER_RASTER["rows"], INFER_RASTER["cols"], 3, tile_size=256, seed=5,
                                      areas=INFER_RASTER["areas"], shard=(a.rank, a.world), shard_mode=a.shard_mode)
            out = predict_linear_blend(DataLoader(ds, batch_size=5, shuffle=False), model)      # reduce_to_rank0 default
            torch.save({"raster": torch.from_numpy(out.copy()), "n_tiles": len(ds)}, a.out)
        elif a.mode == "trainer":
            run_trainer(a, dev)
        else:
            raise SystemExit(f"unknown mode {a.mode}")
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
