#!/usr/bin/env python3
"""ResDepth hot-path benchmark: DSM tiles/s, forward + backward + Adam (+ DP gradient sync) on MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[1] (cfg-S): config_ResDepth-stereo, 3-channel 256x256 tiles, depth-5 U-Net,
batch 32 per GPU (weak scaling: 8 GPUs = global batch 256), fp32, synthetic tiles already resident in HBM,
default-initialised weights.  One "step" = the reference's training iteration (lib/Trainer.py:159-222):
forward, masked de-normalised L1, backward, Adam step, gradients cleared -- with the scalar loss kept on the
device and read back after the timed region (the reference's per-step loss.item() would only add a host sync).

Prints ONE JSON line (rank 0).  `roofline` describes the dominant kernel class, timed with HIP events on the
launch stream over the timed region (rd_prof_*, include/resdepth_hip.h); `cpu_baseline` is the oracle
(torch-CPU restatement of the reference step) on a bounded sample on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FLOP_PER_TILE = 59.33e9        # fwd+bwd, cfg-S (SURVEY.md 8d / BASELINE.md section 2)
PEAK_F32_TFLOPS = 157.3        # MI355X_MICROARCH.md: f32-input MFMA = fp32 vector peak
PEAK_BF16_TFLOPS = 2500.0      # MI355X_MICROARCH.md: dense bf16 MFMA (2495 measured)
# split-bf16 kernels (DESIGN.md "split-bf16 MFMA"): every fp32 multiply-add costs six bf16 MFMA products, so the matrix
# pipe bounds them at 2500/6 fp32-equivalent TFLOP/s
PEAK_SPLIT_TFLOPS = PEAK_BF16_TFLOPS / 6.0
PEAK_HBM_GBS = 8000.0
MFMA_CLASSES = ("conv3x3_fwd", "conv3x3_dgrad", "conv3x3_wgrad", "convt2x2_fwd", "convt2x2_dgrad", "convt2x2_wgrad")


def cpu_baseline(batch=4, timed=2):
    """The oracle's train step (same module graph / loss / Adam as the reference) on the host CPU."""
    from oracle import unet_oracle as O
    # thread count: best of a measured sweep on the 2 x 64-core EPYC 9575F host of the MI355X box
    # (8: 0.59, 16: 0.54, 32: 0.62, 64: 1.10 s/step at batch 4; all 256 hardware threads: 62 s/step)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    spec = O.Spec(n_input_channels=3, start_kernel=64, depth=5, bias_conv_layer=True)
    sd = O.init_state_dict(spec, 0)
    b = O.synthetic_batch(batch, 3, 256, seed=1234)
    state = {}
    O.train_step(sd, b, spec, state)                      # warm-up
    t0 = time.perf_counter()
    for _ in range(timed):
        O.train_step(sd, b, spec, state)
    dt = (time.perf_counter() - t0) / timed
    return {"value": round(batch / dt, 3), "unit": "tiles/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{timed} timed + 1 warm-up train steps (fwd+bwd+Adam) of batch {batch}, 3-ch 256x256 depth-5, "
                      f"torch-CPU oracle, {dt * 1e3:.0f} ms/step"}


def infer_main(args, world, rank, dev):
    """cfg-G: forward-only sweep of a raster with overlapping tiles + linear blend (lib/evaluation.py:460-513).
    `steps` = number of full sweeps timed.  Tiles are sharded round-robin over the ranks; rasters are summed on rank 0."""
    from torch.utils.data import DataLoader
    from resdepth_amd import UNet, SyntheticRasterTiles, predict_linear_blend, _lib
    if world > 1:
        import torch.distributed as dist
        # no device_id=: with the eager communicator initialisation it triggers, every step of this process ran 1.5-2 ms
        # slower on the MI355X boxes (with or without collectives in flight); torch.cuda.set_device above already pins
        # the rank to its GPU for the lazily created communicator
        dist.init_process_group("nccl")
    torch.manual_seed(0)
    model = UNet(n_input_channels=3, start_kernel=64, depth=5, bias_conv_layer=True).to(dev).eval()
    ds = SyntheticRasterTiles(args.raster, args.raster, 3, tile_size=256, seed=1, shard=(rank, world))
    # tiles are staged on the device once (the metric excludes host->device staging, as for training)
    batches = []
    for b in DataLoader(ds, batch_size=args.batch, shuffle=False):
        batches.append({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()})

    class Loader(list):
        dataset = ds
    loader = Loader(batches)
    n_tiles_global = len(SyntheticRasterTiles(args.raster, args.raster, 3, tile_size=256, seed=1).pos)
    for _ in range(max(1, args.warmup)):
        predict_linear_blend(loader, model)
    torch.cuda.synchronize()
    if not args.no_prof:
        _lib.prof_reset(); _lib.prof_enable(2 if args.prof_all else 1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = predict_linear_blend(loader, model)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kern = []
    if not args.no_prof:
        _lib.prof_enable(False); kern = _lib.prof_collect()
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    if rank == 0:
        tiles_s = n_tiles_global * args.steps / dt
        fwd_flop = 19.80e9
        print(json.dumps({
            "metric": "DSM tiles/sec forward-only tiled inference + linear blend (256x256, 3-ch, depth-5 U-Net)",
            "value": round(tiles_s, 2), "unit": "tiles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic raster",
            "config": {"workload": f"cfg-G: {args.raster}x{args.raster} raster, {n_tiles_global} tiles of 256x256 at stride 128, "
                                   f"eval-mode BN, batch {args.batch}", "parallelism": f"tiles sharded over {world} GPU(s)"},
            "e2e": {"tflops": round(tiles_s / world * fwd_flop / 1e12, 2),
                    "frac_f32_peak": round(tiles_s / world * fwd_flop / 1e12 / PEAK_F32_TFLOPS, 4)},
            "raster_checksum": float(out.sum()),
            "kernels": [{"name": k["name"], "ms_per_sweep": round(k["ms"] / args.steps, 3)} for k in
                        sorted(kern, key=lambda e: -e["ms"])[:8]]}), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="tiles per GPU")
    ap.add_argument("--sync-bn", action="store_true", help="SyncBN (single-device-equivalent statistics)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="disable the per-kernel HIP-event timing")
    ap.add_argument("--serial-backward", action="store_true",
                    help="disable the two-stream backward (weight gradients overlapping the dgrad/BN chain) in the timed region")
    ap.add_argument("--prof-steps", type=int, default=5, help="steps of the serialized, instrumented roofline pass")
    ap.add_argument("--prof-all", action="store_true",
                    help="bracket EVERY kernel launch with HIP events (complete breakdown; costs ~4%% of the step). Default: "
                         "only the MFMA kernel classes the roofline needs (~1%%)")
    ap.add_argument("--workload", choices=["S", "M"], default="S",
                    help="S = cfg-S (BASELINE configs[1], the headline metric); M = cfg-M (configs[3]: 2-ch 512x512 depth-6)")
    ap.add_argument("--from-rasters", action="store_true",
                    help="draw a fresh augmented batch from HBM-resident synthetic rasters with resdepth_amd.GpuPatchSampler "
                         "inside every timed step (sample assembly + train step) instead of re-using one resident batch")
    ap.add_argument("--infer", action="store_true",
                    help="measure the tiled full-raster inference sweep instead (BASELINE configs[4], cfg-G: 3-ch tiles of "
                         "256x256 at stride 128 over a synthetic --raster x --raster DSM, eval-mode BN, linear blend)")
    ap.add_argument("--raster", type=int, default=4096)
    ap.add_argument("--force-dist", action="store_true",
                    help="run the data-parallel code path (RCCL process group, bucketed all-reduce) even at world size 1")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus} "
              f"(WORLD_SIZE={world})", file=sys.stderr)
        sys.exit(2)
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from resdepth_amd import UNet, FusedAdam, masked_l1_loss, _lib, dp
    from oracle import unet_oracle as O   # synthetic batch generator + cpu_baseline only (never the measured path)
    _lib.load()

    if args.infer:
        return infer_main(args, world, rank, dev)
    gs = None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:          # --force-dist without a launcher
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        # no device_id=: with the eager communicator initialisation it triggers, every step of this process ran 1.5-2 ms
        # slower on the MI355X boxes (with or without collectives in flight); torch.cuda.set_device above already pins
        # the rank to its GPU for the lazily created communicator
        dist.init_process_group("nccl")
    torch.manual_seed(0)
    wl = {"S": dict(c=3, t=256, depth=5, flop=FLOP_PER_TILE, name="config_ResDepth-stereo (cfg-S): 3-ch 256x256 tiles, depth-5 U-Net"),
          "M": dict(c=2, t=512, depth=6, flop=241.66e9, name="config_ResDepth-mono (cfg-M): 2-ch 512x512 tiles, depth-6 U-Net")}[args.workload]
    model = UNet(n_input_channels=wl["c"], start_kernel=64, depth=wl["depth"], bias_conv_layer=True).to(dev).train()
    if use_dist:
        gs = dp.attach(model, sync_bn=args.sync_bn)
        dp.broadcast_parameters(model, 0)
    opt = FusedAdam(model.parameters(), lr=2e-4, weight_decay=1e-5)
    model.two_stream_backward = not args.serial_backward

    n = args.batch
    b = O.synthetic_batch(n, wl["c"], wl["t"], seed=1234 + rank)
    x = b["input"].to(dev)
    y = b["target"].to(dev)
    mask = b["loss_mask"].to(dev)
    mean, std = b["dsm_mean"].to(torch.float32).to(dev), b["dsm_std"].to(dev)
    params = list(model.parameters())
    losses = []
    sampler = None
    if args.from_rasters:
        from resdepth_amd import GpuPatchSampler
        gr = torch.Generator().manual_seed(99 + rank)
        R = 4096
        dsm_r = torch.randn(R, R, generator=gr) * 3.0 + 400.0
        sampler = GpuPatchSampler(dsm_r, dsm_r + torch.randn(R, R, generator=gr), torch.rand(wl["c"] - 1, R, R, generator=gr) * 200,
                                  tile_size=wl["t"], dsm_std=3.0, ortho_mean=100.0, ortho_std=50.0, device=dev)
        pair = list(range(wl["c"] - 1))

    def step():
        if sampler is not None:
            bb = sampler.random_batch(n, pair, generator=gr)
            xx, yy, mm, me, sd_ = bb["input"], bb["target"], bb["loss_mask"], bb["dsm_mean"], bb["dsm_std"]
        else:
            xx, yy, mm, me, sd_ = x, y, mask, mean, std
        y_pred = model(xx)
        loss = masked_l1_loss(y_pred, yy, mm, me, sd_, grad_sync=gs)
        loss.backward()
        opt.step()
        for p in params:
            p.grad = None                     # lib/Trainer.py:221-222
        losses.append(loss.detach())

    def barrier():
        if use_dist:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    losses.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    timed_losses = list(losses)
    # ---- per-kernel roofline pass.  In the timed region the weight-gradient kernels run concurrently with the
    # dgrad/BN chain on a second stream, so a kernel's event-to-event duration there includes time shared with another
    # kernel; the roofline numbers therefore come from a SERIALIZED pass of the same step, in this process, right after
    # the timed region (HIP events on the launch stream, rd_prof_*).  `value` is never taken from this pass.
    kern, prof_steps = [], 0
    if not args.no_prof:
        model.two_stream_backward = False
        step()
        torch.cuda.synchronize()
        _lib.prof_reset()
        _lib.prof_enable(2 if args.prof_all else 1)
        prof_steps = max(1, args.prof_steps)
        for _ in range(prof_steps):
            step()
        torch.cuda.synchronize()
        _lib.prof_enable(0)
        kern = _lib.prof_collect()
    losses[:] = timed_losses
    loss_vals = [float(v) for v in losses]
    if use_dist:
        import torch.distributed as dist
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)

    if rank == 0:
        ms = dt / args.steps * 1e3
        tiles_s = n * world * args.steps / dt
        kernels = []
        for k in sorted(kern, key=lambda e: -e["ms"]):
            if k["launches"] == 0:
                continue
            op, _, sym = k["name"].partition("|")
            e = {"name": op, "launches_per_step": k["launches"] / prof_steps,
                 "ms_per_step": round(k["ms"] / prof_steps, 4)}
            if sym:
                e["kernel"] = sym
            if k["flops"] > 0:
                e["tflops"] = round(k["flops"] / (k["ms"] * 1e-3) / 1e12, 2)
            if k["bytes"] > 0:
                e["alg_gbs"] = round(k["bytes"] / (k["ms"] * 1e-3) / 1e9, 1)
            kernels.append(e)
        # roofline of the dominant KERNEL (= one kernel symbol as rocprofv3 reports it; e.g. the conv3x3 forward
        # and data-gradient launches are the same igemm_nt instantiation)
        roof = None
        by_sym = {}
        for k in kern:
            op, _, sym = k["name"].partition("|")
            if op in MFMA_CLASSES and k["launches"] > 0:
                g = by_sym.setdefault(sym or op, {"ms": 0.0, "flops": 0.0, "launches": 0, "ops": set()})
                g["ms"] += k["ms"]; g["flops"] += k["flops"]; g["launches"] += k["launches"]; g["ops"].add(op)
        if by_sym:
            sym, dom = max(by_sym.items(), key=lambda kv: kv[1]["ms"])
            ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
            traffic = None
            try:        # HBM bytes per launch from the committed rocprofv3 PMC passes (scripts/summarize_prof.py)
                with open(os.path.join(ROOT, "profiles", "r01_summary.json")) as f:
                    traffic = json.load(f)["kernels"][sym]["hbm_bytes_per_launch"]
            except Exception:
                traffic = None
            is_split = any(tag in sym for tag in ("split", "strip"))
            peak = PEAK_SPLIT_TFLOPS if is_split else PEAK_F32_TFLOPS
            roof = {"kernel": sym, "ops": sorted(dom["ops"]), "bound": "mfma", "achieved": round(ach, 2),
                    "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                    "peak_note": ("fp32-equivalent FLOP/s; bound = dense bf16 MFMA peak (2500 TF) / 6 products per fp32 "
                                  "multiply-add of the exact 3-term split" if is_split else "f32-input MFMA peak"),
                    "frac_of_f32_mfma_peak": round(ach / PEAK_F32_TFLOPS, 4),
                    "traffic": traffic, "avg_launch_ms": round(dom["ms"] / dom["launches"], 4),
                    "alg_flop_per_launch": dom["flops"] / dom["launches"],
                    "launches_per_step": dom["launches"] / prof_steps,
                    "measured": f"HIP events, serialized pass of {prof_steps} steps right after the timed region "
                                "(timed region itself: un-instrumented, wgrad kernels overlapped on a 2nd stream)"}
        out = {
            "metric": "DSM tiles/sec fwd+bwd (256x256, 3-ch, depth-5 U-Net)" if args.workload == "S" else
                      "DSM tiles/sec fwd+bwd (512x512, 2-ch, depth-6 U-Net)", "value": round(tiles_s, 2),
            "unit": "tiles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32",
            "arithmetic": ("fp32 storage and accumulation; MFMA-class kernels multiply exactly-split operands "
                           "(x = x1+x2+x3, bf16 terms) on v_mfma_f32_32x32x16_bf16, 6 products per multiply, error below one fp32 "
                           "rounding -- same parity tolerances as the exact-f32 MFMA path (RD_MFMA=f32)"),
            "data": ("synthetic rasters resident in HBM, a fresh augmented batch assembled on the GPU every step"
                                     if args.from_rasters else
                                     "synthetic (randn tiles resident in HBM, default-initialised weights)"),
            "config": {"workload": wl["name"] + ", fwd+loss+bwd+Adam",
                       "tiles_per_gpu": n, "global_batch": n * world,
                       "parallelism": f"dp{world}" + ("+syncbn" if args.sync_bn else ""),
                       "backward": "serial" if args.serial_backward else "two-stream (wgrad || dgrad+BN)"},
            "e2e": {"tflops": round(tiles_s / world * wl["flop"] / 1e12, 2),
                    "frac_f32_peak": round(tiles_s / world * wl["flop"] / 1e12 / PEAK_F32_TFLOPS, 4)},
            "roofline": roof,
            "kernels": kernels,
            "loss_first_last": [round(loss_vals[0], 6), round(loss_vals[-1], 6)] if loss_vals else None,
        }
        if world == 1 and not args.no_cpu_baseline and args.workload == "S":
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if use_dist:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
