#!/usr/bin/env python3
"""ResDepth hot-path benchmark: DSM tiles/s, forward + backward + Adam (+ DP gradient sync) on MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N --steps K --warmup W        # no launcher: bench.py starts its N ranks itself (self_launch)

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
import subprocess
import sys
import time

# RCCL on these hosts: the kernel driver supports only dmabuf IPC handles; with the legacy IPC mode RCCL's intra-node
# peer-to-peer set-up (and any sharing of device tensors between processes) fails with `hipIpcGetMemHandle: invalid
# argument`.  The HSA runtime reads the variable when it initialises, i.e. at the first HIP call of the process -- so it is
# set here, before torch is imported, and handed to the ranks self_launch() starts.  The image exports it already; the
# default only matters for environments built by hand (a bare `env -i python bench.py --gpus 8`).
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FLOP_PER_TILE = 59.33e9        # fwd+bwd, cfg-S (SURVEY.md 8d / BASELINE.md section 2)
PEAK_F32_TFLOPS = 157.3        # MI355X_MICROARCH.md: f32-input MFMA = fp32 vector peak
PEAK_BF16_TFLOPS = 2500.0      # MI355X_MICROARCH.md: dense bf16 MFMA (2495 measured)
# split kernels: every fp32 multiply-add costs `products` 16-bit MFMA products -- 3 in the default split2h
# arithmetic (two fp16 terms), 6 in split3 (three bf16 terms) -- so the matrix pipe bounds them at 2500 / products
# fp32-equivalent TFLOP/s
def peak_split_tflops(products):
    return PEAK_BF16_TFLOPS / float(products)


PEAK_SPLIT_TFLOPS = PEAK_BF16_TFLOPS / 6.0
PEAK_HBM_GBS = 8000.0
# measured bounds: scripts/split_numerics.py / tests/test_split_numerics_gpu.py
# measured bounds: tests/test_split2h_gpu.py, tests/test_split_numerics_gpu.py
ARITHMETIC_SPLIT2H = ("fp32 storage and accumulation; MFMA-class kernels multiply every operand as TWO fp16 terms of s*x (s = a power of "
                      "two per operand tensor from the tensor's maximum, which the producing kernel's epilogue leaves in a device slot: "
                      "order-independent integer max, bit-reproducible) with THREE products per multiply (a1 b1 | a1 b2, a2 b1 in a "
                      "second fp32 accumulator) on v_mfma_f32_32x32x16_f16, scaled back exactly: <= 3 * 2^-22 |ab| per product; per op "
                      "against fp64 the error is BELOW the exact-f32 MFMA chain's on every operand flavour (0.17-0.23 u rms vs 0.34-0.47 "
                      "on N(0,1) data at K = 4608, u = 2^-24; tests/test_split2h_gpu.py), every whole-net parity bar of tests/ holds; a "
                      "launch whose operand has no slot or an infinite element runs the six-product split3 body (decided on the device): "
                      "fp32's non-finite semantics.  Inference keeps split3 (results independent of the batch a tile is in). "
                      "RD_MFMA=split3 / f32 select the other arithmetics")
ARITHMETIC_SPLIT3 = ("RD_MFMA=split3: fp32 storage and accumulation; MFMA-class kernels multiply exactly-split operands (x = x1+x2+x3, bf16 "
                     "terms by truncation) on v_mfma_f32_32x32x16_bf16, 6 of the 9 term products per multiply -- the three dropped products "
                     "are below 2^-21 |ab| (worst case); hi / lo fp32 accumulators merged once per output; +-Inf / NaN operands propagate "
                     "exactly like fp32 in the forward / data-gradient kernels")
ARITHMETIC = ARITHMETIC_SPLIT2H
MFMA_RANDOM_OPERAND_TFLOPS = 1850.0     # measured, scripts/ubench/mfma_order.hip: register-only bf16 MFMA loop on split terms (hi/mid/lo) of N(0,1) floats
MFMA_F16_SPLIT_OPERAND_TFLOPS = 1717.0  # measured r06, scripts/ubench/mfma_f16.hip: the same loop on the two fp16 terms of scaled N(0,1) floats (profiles/r06_mfma_f16.txt)
MFMA_CLASSES = ("conv3x3_fwd", "conv3x3_dgrad", "conv3x3_wgrad", "convt2x2_fwd", "convt2x2_dgrad", "convt2x2_wgrad")


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _parse_cpulist(text):
    """'0-15,128-143' -> [0..15, 128..143] (the format of /sys/.../local_cpulist)."""
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.extend(range(int(a), int(b or a) + 1))
    return cpus


def _gpu_local_cpus(index):
    """CPUs of the NUMA node HIP device `index` hangs off, from the PCI address torch reports for it:
    /sys/bus/pci/devices/<domain:bus:device.0>/{numa_node,local_cpulist}.  -> (numa_node | None, [cpu, ...] | None)."""
    try:
        pr = torch.cuda.get_device_properties(index)
        bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        base = os.path.join("/sys/bus/pci/devices", bdf)
        with open(os.path.join(base, "numa_node")) as f:
            node = int(f.read())
        with open(os.path.join(base, "local_cpulist")) as f:
            cpus = _parse_cpulist(f.read())
        return node, (cpus or None)
    except Exception:       # noqa: BLE001 -- containers without sysfs PCI topology, older torch: no pinning
        return None, None


def _core_groups(cpus):
    """Group hardware threads into physical cores (thread_siblings_list), keeping the order of first appearance."""
    seen, groups = set(), []
    allowed = set(cpus)
    for c in cpus:
        if c in seen:
            continue
        try:
            with open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list") as f:
                sib = [x for x in _parse_cpulist(f.read()) if x in allowed]
        except Exception:   # noqa: BLE001
            sib = [c]
        sib = sib or [c]
        groups.append(sib)
        seen.update(sib)
    return groups


def plan_rank_affinity(n_local, device_of_rank, allowed):
    """CPU slice of EVERY local rank (pure function of the sysfs topology, so every rank computes the same plan):
    rank -> [cpu, ...], the CPUs of its GPU's NUMA node that this process may use, with the physical cores of a node split
    evenly between the ranks whose GPUs hang off it (both hardware threads of a core stay together).  -> (plan, reason): plan is
    None -- nobody pins -- unless EVERY rank gets at least two physical cores: a mix of pinned and floating ranks would put
    the floating ones on the pinned ones' cores."""
    allowed_set = set(allowed)
    local = {}
    for r in range(n_local):
        node, cpus = _gpu_local_cpus(device_of_rank(r))
        cpus = [c for c in (cpus or []) if c in allowed_set]
        if not cpus:
            return None, f"rank {r}: no local_cpulist for its device's PCI address inside this process's CPU mask"
        local[r] = (node, tuple(cpus))
    if len({v[1] for v in local.values()}) == 1 and len(local[0][1]) == len(allowed) and all(v[0] is None or v[0] < 0 for v in local.values()):
        return None, "the devices report no NUMA locality (numa_node -1, every CPU local)"
    plan = {}
    for key in {v[1] for v in local.values()}:
        sharers = sorted(r for r, v in local.items() if v[1] == key)
        cores = _core_groups(list(key))
        per = len(cores) // len(sharers)
        if per < 2:
            return None, f"{len(cores)} physical cores for the {len(sharers)} ranks of one NUMA node"
        for k, r in enumerate(sharers):
            plan[r] = [c for grp in cores[k * per:(k + 1) * per] for c in grp]
    return plan, None


def pin_rank_to_gpu_numa(local_rank, n_local, device_index):
    """Bind this rank (its launch thread and every thread it starts later: OpenMP, the autograd worker) to the CPUs
    `plan_rank_affinity` gives it: eight launch threads then neither float across the two sockets nor share cores.  Falls back
    silently -- and says why in the returned record -- when the topology is not visible or not every rank can be pinned.
    -> dict for the `dist.affinity_per_rank` field."""
    info = {"pinned": False}
    try:
        allowed = sorted(os.sched_getaffinity(0))
        info["cpus_before"] = len(allowed)
        # --share-gpu (all ranks on one device) maps every rank to that device; otherwise rank r drives GPU r
        dev_of = (lambda r: r) if device_index == local_rank else (lambda r: device_index)
        plan, why = plan_rank_affinity(n_local, dev_of, allowed)
        info["numa_node"] = _gpu_local_cpus(device_index)[0]
        if plan is None:
            info["reason"] = why
            return info
        mine = plan[local_rank]
        os.sched_setaffinity(0, mine)
        node_cpus = set(_gpu_local_cpus(device_index)[1] or [])
        info.update(pinned=True, cpus=len(mine), cpu_first_last=[min(mine), max(mine)],
                    ranks_on_node=sum(1 for cpus in plan.values() if node_cpus.intersection(cpus)))
    except Exception as e:  # noqa: BLE001 -- never take the benchmark down for an affinity call
        info["reason"] = repr(e)[:120]
    return info


def cpu_baseline(warm=1, timed=3, batch=32):
    """BASELINE.md section 4 / SURVEY 8d: the oracle's train step (same module graph / loss / Adam as the reference; the
    oracle is the CPU restatement pinned on the reference, oracle/unet_oracle.py) on this box's host cores -- cfg-S (3-ch)
    at the GPU leg's batch 32, 1 warm-up + 3 timed steps, median (about 20 s of CPU work), and cfg-0 (1-ch, batch 4,
    BASELINE configs[0]) beside it."""
    from oracle import unet_oracle as O
    host = os.cpu_count() or 1

    def run(c, batch, warm, timed):
        spec = O.Spec(n_input_channels=c, start_kernel=64, depth=5, bias_conv_layer=True)
        sd = O.init_state_dict(spec, 0)
        b = O.synthetic_batch(batch, c, 256, seed=1234)
        state, ts = {}, []
        for i in range(warm + timed):
            t0 = time.perf_counter()
            O.train_step(sd, b, spec, state)
            if i >= warm:
                ts.append(time.perf_counter() - t0)
        ts.sort()
        med = ts[len(ts) // 2]
        return {"tiles_per_s": round(batch / med, 3), "batch": batch, "step_s_median": round(med, 4),
                "step_s_min_max": [round(ts[0], 4), round(ts[-1], 4)]}

    # thread count: swept IN THIS RUN (1 warm-up + 2 timed cfg-S steps at batch 8 per setting), the sample below runs at the
    # best one.  BASELINE.md's "all cores" is not the fast setting on the 2 x 64-core EPYC 9575F hosts of the MI355X boxes:
    # with all 256 hardware threads one step took 62 s in round 1 (oneDNN's per-thread work gets too small and the two
    # sockets thrash), so the sweep stops at 64
    sweep = {}
    for th in (8, 16, 32, 64):
        if th <= host:
            torch.set_num_threads(th)
            sweep[th] = run(3, 8, 1, 2)["tiles_per_s"]
    # the batch-8 sweep only shortlists: per tile the CPU is slower at batch 32 (4.4 GB of saved activations, two sockets) and
    # the best thread count can differ, so the batch-32 sample runs at the two best settings of the sweep and the better one counts
    ranked = sorted(sweep, key=sweep.get, reverse=True)[:2] if sweep else [min(16, host)]
    tried = {}
    for th in ranked:
        torch.set_num_threads(th)
        tried[th] = run(3, batch, warm, timed)
    threads = max(tried, key=lambda th: tried[th]["tiles_per_s"])
    torch.set_num_threads(threads)
    big = tried[threads]
    c0 = run(1, 4, 2, 5)
    return {"value": big["tiles_per_s"], "unit": "tiles/s", "cores": threads, "kind": "port",
            "host_cores": host, "threads": threads, "cpu_model": _cpu_model(),
            "cfg_S": big, "cfg_0": c0, "thread_sweep_tiles_per_s_at_batch_8": {str(k): v for k, v in sweep.items()},
            "cfg_S_tiles_per_s_by_threads": {str(k): v["tiles_per_s"] for k, v in tried.items()},
            "sample": f"{warm} warm-up + {timed} timed train steps (fwd+loss+bwd+Adam), median: cfg-S 3-ch 256x256 depth-5 at batch "
                      f"{big['batch']} -- the GPU leg's batch -- ({big['step_s_median'] * 1e3:.0f} ms/step), and cfg-0 1-ch at batch "
                      f"{c0['batch']} ({c0['step_s_median'] * 1e3:.0f} ms/step); torch-CPU oracle, {threads} threads of {host} "
                      f"hardware threads (thread count: the better of the two best settings of an in-run sweep at batch 8)"}


def infer_main(args, world, rank, dev):
    """cfg-G: forward-only sweep of a raster with overlapping tiles + linear blend (lib/evaluation.py:460-513).
    `steps` = number of full sweeps timed.  Tiles are sharded over the ranks by row bands (resdepth_amd.tiling.band_shards):
    band-sized private rasters, the shared rows exchanged point to point, every rank's rows copied into one shared host array."""
    from torch.utils.data import DataLoader
    from resdepth_amd import UNet, SyntheticRasterTiles, predict_linear_blend, _lib
    use_dist = world > 1 or args.force_dist          # --force-dist: the RCCL reduce of the sweep at world size 1
    if use_dist:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:          # --force-dist without a launcher
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29534", RANK="0", WORLD_SIZE="1")
        init_dist(args)
    torch.manual_seed(0)
    model = UNet(n_input_channels=3, start_kernel=64, depth=5, bias_conv_layer=True).to(dev).eval()
    model.fold_eval_bn = not args.no_fold
    model.fast_eval = bool(args.fast_eval)       # one scale per TENSOR in the sweep (results then depend on the batch a tile is in)
    model.eval_per_image = not args.eval_six     # default: three-product bodies with one scale per IMAGE (a tile depends on itself alone)
    model_fast_eval = (model.fast_eval or model.eval_per_image) and _lib.products() == 3
    eval_scales = None if not model_fast_eval else "one per tensor (fast_eval)" if model.fast_eval else "one per image (rd_quant_next_img)"
    ds = SyntheticRasterTiles(args.raster, args.raster, 3, tile_size=256, seed=1, shard=(rank, world))
    # tiles are staged on the device once (the metric excludes host->device staging, as for training)
    batches = []
    for b in DataLoader(ds, batch_size=args.batch, shuffle=False):
        batches.append({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()})

    class Loader(list):
        dataset = ds
    loader = Loader(batches)
    n_tiles_global = ds.shard_plan[-1]["i1"]
    for _ in range(max(1, args.warmup)):
        predict_linear_blend(loader, model, host="reuse")
    torch.cuda.synchronize()
    if use_dist:
        import torch.distributed as dist
        dist.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = predict_linear_blend(loader, model, host="reuse")      # same pinned / shared host array every sweep
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    checksum = float(out.sum())
    kern = []
    if not args.no_prof:        # one more sweep, instrumented with HIP events per kernel class (never part of `value`)
        _lib.prof_reset(); _lib.prof_enable(2)
        predict_linear_blend(loader, model, reduce_to_rank0=False)
        torch.cuda.synchronize()
        _lib.prof_enable(False); kern = _lib.prof_collect()
    dist_info = None
    if use_dist:
        import torch.distributed as dist
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
        dist_info = {"backend": dist.get_backend(), "world_size_reported": dist.get_world_size()}
    record = None
    me = ds.shard_plan[rank]
    sweep_info = {"sharding": "row bands (tiling.band_shards)", "tiles_per_rank": [p["i1"] - p["i0"] for p in ds.shard_plan],
                  "private_raster_rows_rank0": me["hi"] - me["lo"], "owned_rows_per_rank": [p["c1"] - p["c0"] for p in ds.shard_plan],
                  "exchanged_mbytes_per_boundary": round((256 - 128) * args.raster * 8 / 2 ** 20, 2) if world > 1 else 0.0,
                  "host_array": "shared memory, page-locked per rank" if use_dist else "pinned",
                  "d2h": "row stripes on a copy stream during the sweep"}
    if rank == 0:
        tiles_s = n_tiles_global * args.steps / dt
        fwd_flop = 19.80e9
        record = {
            "metric": "DSM tiles/sec forward-only tiled inference + linear blend (256x256, 3-ch, depth-5 U-Net)",
            "value": round(tiles_s, 2), "unit": "tiles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "arithmetic_mode": ("split2h" if model_fast_eval else "split3") if _lib.mfma_mode() != "f32" else "f32",
            "eval_scales": eval_scales,
            "data": "synthetic raster",
            "config": {"workload": f"cfg-G: {args.raster}x{args.raster} raster, {n_tiles_global} tiles of 256x256 at stride 128, "
                                   f"eval-mode BN, batch {args.batch}",
                       "parallelism": f"tiles sharded over {world} GPU(s)" + (" (ranks SHARE one GPU: code-path check)" if args.share_gpu else "")},
            "e2e": {"tflops": round(tiles_s / world * fwd_flop / 1e12, 2),
                    "frac_f32_peak": round(tiles_s / world * fwd_flop / 1e12 / PEAK_F32_TFLOPS, 4)},
            "roofline": build_roofline(kern, 1, PMC_SUMMARIES["G"], products=3 if model_fast_eval else 6,
                                       mode=("split2h" if model_fast_eval else "split3") if _lib.mfma_mode() != "f32" else "f32",
                                       measured="HIP events, one instrumented sweep right after the timed "
                                       "sweeps (this rank's shard of the tiles)", per="sweep"),
            "raster_checksum": checksum, "dist": dist_info, "sweep": sweep_info,
            "kernels": kernel_rows(kern, 1, per="sweep")}
    emit_line(record, use_dist)


WORKLOADS = {
    "S": dict(c=3, t=256, depth=5, flop=FLOP_PER_TILE, bytes_a=1.123e9,
              name="config_ResDepth-stereo (cfg-S): 3-ch 256x256 tiles, depth-5 U-Net"),
    "M": dict(c=2, t=512, depth=6, flop=241.66e9, bytes_a=4.470e9,
              name="config_ResDepth-mono (cfg-M): 2-ch 512x512 tiles, depth-6 U-Net"),
}


class TrainBench:
    """Model + optimizer + one resident synthetic batch of a workload; step() = the reference's training iteration."""

    def __init__(self, wl, n, dev, rank=0, gs=None, from_rasters=False):
        from resdepth_amd import UNet, FusedAdam, synthetic_batch
        self.wl, self.n, self.gs = wl, n, gs
        self.graph, self.use_graph = None, False      # resdepth_amd.graph.GraphedTrainStep (attach_optimizer): --graph
        self.plan, self.use_plan = None, False        # resdepth_amd.plan.PlannedTrainStep: the default iteration of the headline
        torch.manual_seed(0)
        self.model = UNet(n_input_channels=wl["c"], start_kernel=64, depth=wl["depth"], bias_conv_layer=True).to(dev).train()
        b = synthetic_batch(n, wl["c"], wl["t"], seed=1234 + rank)
        self.x, self.y, self.mask = b["input"].to(dev), b["target"].to(dev), b["loss_mask"].to(dev)
        self.mean, self.std = b["dsm_mean"].to(torch.float32).to(dev), b["dsm_std"].to(dev)
        self.losses = []
        self.sampler = None
        if from_rasters:
            from resdepth_amd import GpuPatchSampler
            self.gr = torch.Generator().manual_seed(99 + rank)
            R = 4096
            dsm_r = torch.randn(R, R, generator=self.gr) * 3.0 + 400.0
            self.sampler = GpuPatchSampler(dsm_r, dsm_r + torch.randn(R, R, generator=self.gr),
                                           torch.rand(wl["c"] - 1, R, R, generator=self.gr) * 200, tile_size=wl["t"], dsm_std=3.0,
                                           ortho_mean=100.0, ortho_std=50.0, device=dev)
            self.pair = list(range(wl["c"] - 1))
            # batch k + 1 is assembled on the sampler's side stream while batch k trains (GpuPatchSampler.stream_batches)
            self.batch_iter = self.sampler.stream_batches(10 ** 9, n, self.pair, generator=self.gr, prefetch=1)

    def attach_optimizer(self):
        from resdepth_amd import FusedAdam
        self.opt = FusedAdam(self.model.parameters(), lr=2e-4, weight_decay=1e-5)
        self.params = list(self.model.parameters())
        from resdepth_amd.graph import GraphedTrainStep
        from resdepth_amd.plan import PlannedTrainStep
        self.graph = GraphedTrainStep(self.model, self.opt, warmup=2)
        self.plan = PlannedTrainStep(self.model, self.opt, warmup=2)

    def step(self):
        from resdepth_amd import masked_l1_loss
        if self.sampler is not None:
            bb = next(self.batch_iter)
            xx, yy, mm, me, sd_ = bb["input"], bb["target"], bb["loss_mask"], bb["dsm_mean"], bb["dsm_std"]
        else:
            xx, yy, mm, me, sd_ = self.x, self.y, self.mask, self.mean, self.std
        if self.use_plan:
            loss = self.plan(xx, yy, mm, me, sd_)         # the recorded launch list replayed from C (eager while it warms up)
            self.losses.append(loss.clone() if self.plan.why_eager is None else loss)
            return
        if self.use_graph and self.gs is None:
            loss = self.graph(xx, yy, mm, me, sd_)        # one hipGraph launch once captured (eager while it warms up)
            self.losses.append(loss.clone() if self.graph.why_eager is None else loss)
            return
        y_pred = self.model(xx)
        loss = masked_l1_loss(y_pred, yy, mm, me, sd_, grad_sync=self.gs)
        loss.backward()
        self.opt.step()
        for p in self.params:
            p.grad = None                     # lib/Trainer.py:221-222
        self.losses.append(loss.detach())

    def timed(self, steps, warmup, barrier=None):
        """-> (wall seconds for `steps` steps between barriers, per-step HIP-event durations in ms).  The events are
        recorded on torch's current stream, which every step joins with the weight-gradient stream before Adam, so one
        event pair brackets the whole step."""
        barrier = barrier or torch.cuda.synchronize
        for _ in range(warmup):
            self.step()
        barrier()
        self.losses.clear()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        t0 = time.perf_counter()
        ev[0].record()
        for i in range(steps):
            self.step()
            ev[i + 1].record()
        barrier()
        dt = time.perf_counter() - t0
        return dt, [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]


# committed rocprofv3 PMC summaries (scripts/profile.sh + scripts/summarize_prof.py) per workload, newest first: the source
# of `roofline.traffic` / `roofline.pmc` -- counters are never collected by bench.py itself
PMC_SUMMARIES = {"S": ("r06_summary.json",), "M": ("r06_cfgM_summary.json",), "G": ("r06_cfgG_summary.json",)}


def kernel_rows(kern, prof_steps, per="step"):
    """rd_prof_collect() records -> the `kernels` list of the JSON line (per step / per sweep)."""
    rows = []
    for k in sorted(kern, key=lambda e: -e["ms"]):
        if k["launches"] == 0:
            continue
        op, _, sym = k["name"].partition("|")
        e = {"name": op, f"launches_per_{per}": k["launches"] / prof_steps, f"ms_per_{per}": round(k["ms"] / prof_steps, 4)}
        if sym:
            e["kernel"] = sym
        if k["flops"] > 0:
            e["tflops"] = round(k["flops"] / (k["ms"] * 1e-3) / 1e12, 2)
        if k["bytes"] > 0:
            e["alg_gbs"] = round(k["bytes"] / (k["ms"] * 1e-3) / 1e9, 1)
            if op not in MFMA_CLASSES:              # HBM-class kernel: fraction of the 8 TB/s spec on ALGORITHMIC bytes
                e["hbm_frac"] = round(e["alg_gbs"] / PEAK_HBM_GBS, 3)
        rows.append(e)
    return rows


def build_roofline(kern, prof_steps, summaries, measured, per="step", products=None, mode=None):
    """`roofline` of the dominant KERNEL (= one kernel symbol as rocprofv3 reports it; e.g. the conv3x3 forward and
    data-gradient launches are the same conv3_halo_split instantiation): algorithmic FLOP per launch / average launch
    duration from the HIP events of rd_prof_*, against the pipe that bounds it."""
    by_sym = {}
    for k in kern:
        op, _, sym = k["name"].partition("|")
        if op in MFMA_CLASSES and k["launches"] > 0:
            g = by_sym.setdefault(sym or op, {"ms": 0.0, "flops": 0.0, "launches": 0, "ops": set()})
            g["ms"] += k["ms"]; g["flops"] += k["flops"]; g["launches"] += k["launches"]; g["ops"].add(op)
    if not by_sym:
        return None
    sym, dom = max(by_sym.items(), key=lambda kv: kv[1]["ms"])
    ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
    traffic, traffic_src, pmc = None, None, None
    for name in summaries:
        # HBM bytes per launch are NOT measured by this process: they come from the committed rocprofv3 PMC passes
        # of this same command (scripts/profile.sh + scripts/summarize_prof.py; FETCH_SIZE doubled per the guide)
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                whole = json.load(f)
            from resdepth_amd import _lib as _l3
            have = (_l3.load().rd_version(), mode or _l3.mfma_mode())
            if (whole.get("rd_version"), whole.get("arithmetic_mode")) != have:
                # a summary of another library / arithmetic: its bytes and pipe figures describe other kernels under the same
                # symbol names -- no traffic rather than someone else's (r05 verdict, evidence defect 16)
                traffic_src = (f"profiles/{name} was taken with library version {whole.get('rd_version')} / mode "
                               f"{whole.get('arithmetic_mode')}, this run is {have[0]} / {have[1]}: traffic not quoted")
                continue
            rec = whole["kernels"][sym]
            traffic = rec["hbm_bytes_per_launch"]
            # same passes: MFMA pipe busy fraction IN CYCLES and the effective clock (power-limited DVFS) -- the
            # product of the two, relative to 2.4 GHz, is what `frac` sees
            pmc = {"mfma_pipe_busy": rec.get("mfma_pipe_util"), "effective_clock_ghz": rec.get("clock_ghz_under_pmc"),
                   "l2_hit_rate": rec.get("l2_hit_rate"), "source": f"profiles/{name}"}
            traffic_src = f"profiles/{name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py, not this run)"
            break
        except Exception:       # noqa: BLE001
            continue
    is_split = any(tag in sym for tag in ("split", "strip", "convt_"))
    if products is None:
        from resdepth_amd import _lib as _l2
        products = _l2.products()           # 3: split2h (training default), 6: split3
    peak = peak_split_tflops(products) if is_split else PEAK_F32_TFLOPS
    power_peak = (MFMA_F16_SPLIT_OPERAND_TFLOPS if products == 3 else MFMA_RANDOM_OPERAND_TFLOPS) / products
    return {"kernel": sym, "ops": sorted(dom["ops"]), "bound": "mfma", "achieved": round(ach, 2),
            "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(ach / peak, 4),
            "peak_note": (f"fp32-equivalent FLOP/s; bound = dense 16-bit MFMA peak (2500 TF) / {products} products per fp32 "
                          "multiply-add (" + ("two fp16 terms, split2h" if products == 3 else "three bf16 terms, split3") + ")"
                          if is_split else "f32-input MFMA peak"),
            "products_per_multiply": products if is_split else 1,
            "frac_of_f32_mfma_peak": round(ach / PEAK_F32_TFLOPS, 4),
            # scripts/ubench/mfma_power.hip / mfma_order.hip (profiles/r03_notes.md section 9): a register-only loop of
            # back-to-back v_mfma_f32_32x32x16_bf16 (pipe 100 % busy, no memory traffic) sustains 2486 TFLOP/s on constant
            # operands, 1684-1724 on random bits and 1850 on what these kernels feed it (the three split terms of
            # N(0,1) floats) -- the power limit (effective clock 1.78 GHz).  Against THAT ceiling / 6 products:
            # (r06: the fp16 split terms draw more power per MFMA than the bf16 ones -- 1545-1717 vs 1836-1907 TFLOP/s in the same
            # loop, profiles/r06_mfma_f16.txt)
            "power_limited_peak": round(power_peak, 1) if is_split else None,
            "frac_of_power_limited_peak": round(ach / power_peak, 4) if is_split else None,
            "traffic": traffic, "traffic_source": traffic_src, "pmc": pmc, "avg_launch_ms": round(dom["ms"] / dom["launches"], 4),
            "alg_flop_per_launch": dom["flops"] / dom["launches"],
            f"launches_per_{per}": dom["launches"] / prof_steps,
            "measured": measured}


def _median(v):
    v = sorted(v)
    return v[len(v) // 2] if v else None


def infer_sweep(dev, raster, batch, sweeps, warm=1, host_batches=False, fast_eval=False, per_image=True):
    """cfg-G on one GPU: `sweeps` full sweeps of a raster x raster synthetic DSM -> (tiles/s, n_tiles).  Tiles resident in HBM
    (the metric's convention: staging excluded), or host_batches=True: pinned HOST batch dicts, what a DataLoader(pin_memory=True)
    hands lib/evaluation.py:486-498 -- every tile crosses PCIe inside the timed sweep."""
    from torch.utils.data import DataLoader
    from resdepth_amd import UNet, SyntheticRasterTiles, predict_linear_blend
    torch.manual_seed(0)
    model = UNet(n_input_channels=3, start_kernel=64, depth=5, bias_conv_layer=True).to(dev).eval()
    model.fast_eval, model.eval_per_image = bool(fast_eval), bool(per_image)
    ds = SyntheticRasterTiles(raster, raster, 3, tile_size=256, seed=1)
    move = (lambda v: v.pin_memory()) if host_batches else (lambda v: v.to(dev))
    batches = [{k: (move(v) if torch.is_tensor(v) else v) for k, v in b.items()} for b in DataLoader(ds, batch_size=batch, shuffle=False)]

    class Loader(list):
        dataset = ds
    loader = Loader(batches)
    for _ in range(warm):
        predict_linear_blend(loader, model)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(sweeps):
        predict_linear_blend(loader, model)
    torch.cuda.synchronize()
    return len(ds) * sweeps / (time.perf_counter() - t0), len(ds)


class _CyclingLoader:
    """`n_iter` iterations over a few HOST-resident pinned batch dicts (what a DataLoader(pin_memory=True) hands the Trainer,
    train.py:146-161), cycled: every iteration moves a full batch host -> device."""

    def __init__(self, batches, n_iter):
        self.batches, self.n_iter, self.dataset = batches, n_iter, list(range(n_iter * batches[0]["input"].shape[0]))
        self.drop_last = True

    def __len__(self):
        return self.n_iter

    def __iter__(self):
        for i in range(self.n_iter):
            yield self.batches[i % len(self.batches)]


def trainer_loop_measurement(dev, wl, n, iters=24, prefetch=1, launch_plan=False):
    """tiles/s of the drop-in loop itself -- resdepth_amd.Trainer.inference_one_epoch('train') over `iters` host-resident
    pinned batches (lib/Trainer.py:159-222 driven as train.py does) -- next to the bare resident-batch step of `value`."""
    import tempfile
    import types
    from resdepth_amd import UNet, FusedAdam, Trainer, synthetic_batch
    torch.manual_seed(0)
    model = UNet(n_input_channels=wl["c"], start_kernel=64, depth=wl["depth"], bias_conv_layer=True).to(dev).train()
    opt = FusedAdam(model.parameters(), lr=2e-4, weight_decay=1e-5)
    host = []
    for k in range(4):
        b = synthetic_batch(n, wl["c"], wl["t"], seed=4321 + k)
        host.append({k_: (v.pin_memory() if torch.is_tensor(v) else v) for k_, v in b.items()})
    tmp = tempfile.mkdtemp(prefix="rd_bench_trainer_")
    mk = lambda it: _CyclingLoader(host, it)
    a = types.SimpleNamespace(model=model, optimizer=opt, scheduler=None, criterion=torch.nn.L1Loss(reduction="mean"),
                              trainloader=mk(iters), valloader=None, n_epochs=1, evaluate_rate=1, save_model_rate=10 ** 9,
                              freq_average_train_loss=10 ** 9, save_dir=tmp, log_file=None,
                              checkpoint_dir=os.path.join(tmp, "ck"), tboard_log_dir=os.path.join(tmp, "tb"),
                              pretrained_path=None, prefetch_batches=prefetch, launch_plan=launch_plan)
    tr = Trainer(a)
    tr.logger.handlers.clear()
    tr.loader["train"] = mk(8 if launch_plan else 6)
    tr.inference_one_epoch(0, "train")               # warm-up epoch (allocator, pinned staging, packed weights; the plan's recording)
    torch.cuda.synchronize()
    # two timed epochs, the faster one counts: an epoch is ~0.17 s of wall clock, and one host hiccup (a page-locked staging buffer
    # being faulted in, another process on the box) once turned 4390 tiles/s into 1339 in a default run
    dts = []
    for ep in (1, 2):
        tr.loader["train"] = mk(iters)
        t0 = time.perf_counter()
        meters = tr.inference_one_epoch(ep, "train")     # ends with ONE read-back of the epoch's losses
        torch.cuda.synchronize()
        dts.append(time.perf_counter() - t0)
    dt = min(dts)
    del tr, model, opt
    return {"tiles_per_s": round(n * iters / dt, 1), "ms_per_iteration": round(dt / iters * 1e3, 3), "iterations": iters,
            "epochs_timed_ms": [round(v * 1e3, 1) for v in dts],
            "prefetch_batches": prefetch, "loss_avg": round(float(meters["MAE_metric"].avg), 6),
            "h2d_mbytes_per_iteration": round(sum(v.numel() * v.element_size() for v in host[0].values() if torch.is_tensor(v)) / 2 ** 20, 1)}


def eval_stats_measurement(dev, side=8192):
    """resdepth_amd.evaluation.get_statistics (lib/evaluation.py:11-131: masked MAE / RMSE / medians / NMAD) on rasters resident
    in HBM: pixels/s and algorithmic GB/s (one read of the fp64 raster, the fp32 ground truth and the mask per pass: the
    single reduction pass + 8 radix-select passes x 3 medians are what the kernel actually issues)."""
    from resdepth_amd.evaluation import get_statistics
    g = torch.Generator(device="cpu").manual_seed(5)
    gt = (torch.randn(side, side, generator=g) * 3.0 + 400.0).to(dev)
    ras = gt.double() + torch.randn(side, side, generator=g).to(dev).double() * 0.5
    mask = (torch.rand(side, side, generator=g) > 0.05).to(dev)
    get_statistics(ras, gt, -9999.0, mask, None, device=dev)
    torch.cuda.synchronize()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        st = get_statistics(ras, gt, -9999.0, mask, None, device=dev)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    px = side * side
    return {"mpixels_per_s": round(px / dt / 1e6, 1), "ms_per_call": round(dt * 1e3, 3), "raster": f"{side}x{side}",
            "alg_gbs_one_pass": round(px * 13 / dt / 1e9, 1), "MAE": round(st["MAE"], 6), "NMAD": round(st["NMAD"], 6),
            "note": "full statistics incl. exact median / absolute median / NMAD; alg_gbs_one_pass counts ONE read of raster "
                    "(f64) + ground truth (f32) + mask (u8) per call"}


def split3_measurement():
    """RD_MFMA=split3 (six products everywhere) in a child process: the r01-r05 headline arithmetic on the same workloads."""
    env = {k: v for k, v in os.environ.items() if k != "RESDEPTH_HIP_LIB"}
    env["RD_MFMA"] = "split3"
    base = [sys.executable, os.path.abspath(__file__), "--no-cpu-baseline", "--no-secondary", "--no-prof"]
    r = subprocess.run(base + ["--steps", "20", "--warmup", "5"], env=env, capture_output=True, text=True, timeout=600)
    d = json.loads(r.stdout.strip().splitlines()[-1])
    out = {"tiles_per_s": d["value"], "step_ms_median": d["step_ms_median"],
           "note": "python bench.py under RD_MFMA=split3 in a child process, 20 timed steps of the same cfg-S workload: three bf16 "
                   "terms / six products per multiply in every MFMA-class kernel (the default arithmetic of r01-r05)"}
    r = subprocess.run(base + ["--workload", "M", "--steps", "3", "--warmup", "2"], env=env, capture_output=True, text=True, timeout=600)
    out["cfg_M_tiles_per_s"] = json.loads(r.stdout.strip().splitlines()[-1])["value"]
    return out


def graph_small_batch_measurement(dev, wl, n=4, steps=40):
    res = {}
    for mode in ("eager", "graph", "plan"):
        t = TrainBench(wl, n, dev)
        t.attach_optimizer()
        t.use_graph, t.use_plan = mode == "graph", mode == "plan"
        for _ in range(5):
            t.step()
        dt, ev = t.timed(steps, 4)
        res[mode] = {"tiles_per_s": round(n * steps / dt, 1), "step_ms_median": round(_median(ev), 3)}
        if mode == "graph":
            res[mode]["replays"] = t.graph.replays
        if mode == "plan":
            res[mode]["replays"], res[mode]["rejected"] = t.plan.replays, getattr(t.plan, "plan_rejected", None)
        del t
    res["note"] = (f"cfg-S at batch {n}, fwd+loss+bwd+Adam, {steps} timed steps: the eager iteration is bound by its ~110 launches "
                   "(~3.5 ms of host time per step), resdepth_amd.graph.GraphedTrainStep replays it as one HIP graph (0.35 ms of host "
                   "time, bit-identical results); at batch >= 8 the eager iteration wins (no longer launch-bound, and its two-stream "
                   "backward overlaps better than the replayed branches)")
    return res


def secondary_measurements(args, dev, tb):
    """Numbers DESIGN.md quotes beside the headline, measured in the same invocation (few steps each; never `value`)."""
    from resdepth_amd import _lib
    out = {}
    try:        # the exact-f32 MFMA kernels on the same workload (arithmetic A/B of the split-bf16 default)
        _lib.tune_set("mfma_f32", 1)
        tb.model.invalidate_packed()
        dt, ev = tb.timed(5, 2)
        out["exact_f32"] = {"tiles_per_s": round(tb.n * 5 / dt, 1), "step_ms_median": round(_median(ev), 3),
                            "note": "same step on the exact-f32 MFMA kernels (v_mfma_f32_32x32x2_f32), 5 timed steps"}
        # its own roofline: the dominant exact-f32 kernel against the f32-input MFMA peak (serialized instrumented pass)
        two = tb.model.two_stream_backward
        tb.model.two_stream_backward = False
        tb.step()
        torch.cuda.synchronize()
        _lib.prof_reset(); _lib.prof_enable(2)
        for _ in range(2):
            tb.step()
        torch.cuda.synchronize()
        _lib.prof_enable(0)
        kern = _lib.prof_collect()
        tb.model.two_stream_backward = two
        out["exact_f32"]["roofline"] = build_roofline(kern, 2, (), "HIP events, serialized pass of 2 steps in exact-f32 mode")
        out["exact_f32"]["e2e_frac_f32_peak"] = round(tb.n * 5 / dt * tb.wl["flop"] / 1e12 / PEAK_F32_TFLOPS, 4)
    except Exception as e:      # noqa: BLE001 -- a secondary number must never take the headline down
        out["exact_f32"] = dict(out.get("exact_f32") or {}, error=repr(e)[:200])
    finally:
        _lib.tune_set("mfma_f32", 0)
        tb.model.invalidate_packed()
        _lib.prof_enable(0)
    try:        # the six-product arithmetic on the same workloads (the library reads RD_MFMA once, at load time: child run)
        out["split3"] = split3_measurement()
    except Exception as e:      # noqa: BLE001
        out["split3"] = {"error": repr(e)[:300]}
    try:        # the launch-bound regime: the same iteration at batch 4, eager (~110 launches) vs one captured HIP graph per step
        out["hip_graph_small_batch"] = graph_small_batch_measurement(dev, tb.wl)
    except Exception as e:      # noqa: BLE001
        out["hip_graph_small_batch"] = {"error": repr(e)[:300]}
    try:        # the drop-in loop: Trainer.inference_one_epoch over host-resident pinned batches, H2D prefetch on / off
        on = trainer_loop_measurement(dev, tb.wl, tb.n, prefetch=1)
        off = trainer_loop_measurement(dev, tb.wl, tb.n, iters=12, prefetch=0)
        out["trainer_loop"] = dict(on, without_prefetch_tiles_per_s=off["tiles_per_s"],
                                   note="resdepth_amd.Trainer.inference_one_epoch('train'): 4 distinct pinned host batches cycled, "
                                        "every iteration copies its batch host->device (DevicePrefetcher: next batch staged on a "
                                        "copy stream under the current step); deferred loss read-back, FusedAdam; eager enqueue "
                                        "(the Trainer shell's default)")
        try:        # the same loop with Trainer(launch_plan=True): the iteration replayed from its recorded launch plan
            pl = trainer_loop_measurement(dev, tb.wl, tb.n, prefetch=1, launch_plan=True)
            out["trainer_loop"]["launch_plan_tiles_per_s"] = pl["tiles_per_s"]
            out["trainer_loop"]["launch_plan_loss_avg"] = pl["loss_avg"]
        except Exception as e:      # noqa: BLE001
            out["trainer_loop"]["launch_plan_error"] = repr(e)[:200]
    except Exception as e:      # noqa: BLE001
        out["trainer_loop"] = {"error": repr(e)[:300]}
    try:        # sample assembly on the GPU inside every step
        tr = TrainBench(tb.wl, tb.n, dev, from_rasters=True)
        tr.attach_optimizer()
        # the part runs hot by now (clocks drift over a long run): the resident-batch step is re-timed right beside it so that the
        # two numbers are comparable (scripts/from_rasters_probe.py: interleaved, a fresh sampler batch per step costs +0.9 %)
        _, ev_res = tb.timed(10, 3)
        dt, ev = tr.timed(10, 3)
        # the sampler alone (no train step): batches/s of rd_patch_sums + rd_assemble_patches
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            tr.sampler.random_batch(tr.n, tr.pair, generator=tr.gr)
        torch.cuda.synchronize()
        ds = (time.perf_counter() - t0) / 20
        out["from_rasters"] = {"tiles_per_s": round(tr.n * 10 / dt, 1), "step_ms_median": round(_median(ev), 3),
                               "resident_batch_step_ms_median_beside_it": round(_median(ev_res), 3),
                               "sampler_alone_tiles_per_s": round(tr.n / ds, 1), "sampler_alone_ms_per_batch": round(ds * 1e3, 3),
                               "note": "GpuPatchSampler.stream_batches (patch extraction, masked mean centring, normalisation, "
                                       "rot90/flip, loss mask; lib/DsmOrthoDataset.py:161-291; the next batch assembled on a side "
                                       "stream under the current step) from 4096^2 rasters resident in HBM + the train step, 10 "
                                       "timed steps"}
        del tr
    except Exception as e:      # noqa: BLE001
        out["from_rasters"] = {"error": repr(e)[:300]}
    try:
        out["eval_stats"] = eval_stats_measurement(dev)
    except Exception as e:      # noqa: BLE001
        out["eval_stats"] = {"error": repr(e)[:300]}
    torch.cuda.empty_cache()
    try:
        wl = WORKLOADS["M"]
        tm = TrainBench(wl, 32, dev)
        tm.attach_optimizer()
        dt, ev = tm.timed(3, 2)
        ts = 32 * 3 / dt
        out["cfg_M"] = {"tiles_per_s": round(ts, 1), "step_ms_median": round(_median(ev), 3), "tflops": round(ts * wl["flop"] / 1e12, 1),
                        "frac_f32_peak": round(ts * wl["flop"] / 1e12 / PEAK_F32_TFLOPS, 4),
                        "workload": wl["name"] + ", batch 32, fwd+loss+bwd+Adam, 3 timed steps"}
        del tm
    except Exception as e:      # noqa: BLE001
        out["cfg_M"] = {"error": repr(e)[:200]}
    try:
        ts, nt = infer_sweep(dev, 8192, 32, 2)
        out["cfg_G"] = {"tiles_per_s": round(ts, 1), "tiles": nt, "tflops": round(ts * 19.80e9 / 1e12, 1),
                        "workload": "cfg-G on one GPU (SURVEY 8d size): 8192x8192 raster, 3969 tiles of 256x256 at stride 128, "
                                    "eval-mode BN folded, batch 32, forward + linear blend, 2 timed sweeps; three-product split2h "
                                    "arithmetic with one magnitude slot per IMAGE (the inference default since r06: a tile's result "
                                    "does not depend on its batch; the 8 x 8 level and its up-convolution stay on six products)"}
        t6, _ = infer_sweep(dev, 8192, 32, 2, per_image=False)
        out["cfg_G"]["six_product_tiles_per_s"] = round(t6, 1)
        out["cfg_G"]["six_product_note"] = "UNet.eval_per_image = False: the split3 bodies of r01-r05 (same property, twice the matrix instructions)"
        tf, _ = infer_sweep(dev, 8192, 32, 2, fast_eval=True)
        out["cfg_G"]["fast_eval_tiles_per_s"] = round(tf, 1)
        out["cfg_G"]["fast_eval_note"] = ("UNet.fast_eval = True: one scale per operand TENSOR, as in training (results then depend, in "
                                          "the last bits, on which tiles share a batch)")
    except Exception as e:      # noqa: BLE001
        out["cfg_G"] = dict(out.get("cfg_G") or {}, error=repr(e)[:200])
    try:        # the same sweep fed from HOST batches (the drop-in call: a DataLoader's pinned batch dicts), 4096^2 raster
        res, _ = infer_sweep(dev, 4096, 32, 2)
        hst, nt = infer_sweep(dev, 4096, 32, 2, host_batches=True)
        out["cfg_G_host_batches"] = {"tiles_per_s": round(hst, 1), "tiles_per_s_resident": round(res, 1), "tiles": nt,
                                     "h2d_mbytes_per_sweep": round(nt * 3 * 256 * 256 * 4 / 2 ** 20, 1),
                                     "workload": "4096x4096 raster, 961 tiles, batch 32, 2 timed sweeps: pinned host batch dicts (input "
                                                 "staged on a copy stream one batch ahead, offsets / boxes as one pinned block per batch) "
                                                 "vs the same tiles resident in HBM"}
    except Exception as e:      # noqa: BLE001
        out["cfg_G_host_batches"] = {"error": repr(e)[:200]}
    torch.cuda.empty_cache()
    return out


def diagnostics_pass(tb, gs, steps, barrier):
    """A few steps AFTER the timed region with the exchange points bracketed by HIP events (resdepth_amd.dp.Probe) and the
    host side clocked: what the first multi-GPU run needs to be read, not just quoted.  Never part of `value`.
      host_enqueue_ms   wall time of one step() on the launching thread, GPU queue never full (median): must stay well under
                        the step time or the GPU starves (8 ranks x ~110 launches through ctypes on two sockets);
      grad_wait         main stream blocked in GradSync.finish after all of its own work: the EXPOSED gradient all-reduce;
      loss_norm         the blocking two-scalar all-reduce between the loss reduction and its finishing kernel;
      bn_fwd / bn_bwd   SyncBN exchanges (strict-parity mode only);
      bucket_issue_ms_after_step_start   when each gradient bucket's collective was issued on the weight-gradient stream."""
    from resdepth_amd import dp
    probe = dp.Probe()
    if gs is not None:
        gs.probe = probe
    barrier()
    host, evs = [], []
    for _ in range(steps):
        if gs is not None:
            probe.mark_step_start()
        a = torch.cuda.Event(enable_timing=True)
        a.record()
        t0 = time.perf_counter()
        tb.step()
        host.append((time.perf_counter() - t0) * 1e3)
        b = torch.cuda.Event(enable_timing=True)
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    if gs is not None:
        gs.probe = None
    out = {"host_enqueue_ms": round(_median(host), 3), "host_enqueue_ms_min_max": [round(min(host), 3), round(max(host), 3)],
           "step_ms_under_probe": round(_median([a.elapsed_time(b) for a, b in evs]), 3), "probe_steps": steps}
    if gs is not None:
        out.update(probe.summary(steps))
        out["plan"] = gs.describe()
    return out


def dist_tuning_pass(tb, gs, args, barrier, steps):
    """What the FIRST multi-GPU run needs to tune itself (r05 verdict, next 3) -- after the timed region, never part of `value`,
    eager iteration (a launch plan is recorded for ONE bucket plan):
      * bucket sweep: the diagnostics pass again with 4 / 8 / 16 MB gradient buckets -> step time, exposed all-reduce (main stream
        blocked in GradSync.finish), bucket count, when each bucket was issued;
      * CU contention: the MFMA-class kernels' own time per step (HIP events around each launch, rd_prof level 1) WITH the
        collectives in flight and WITHOUT them (grad_sync detached: the same kernels, no RCCL kernels co-resident) -- RCCL's
        channels occupy CUs and LDS the convolution kernels would use;
      * RCCL facts: library version and the channel-count environment (--rccl-channels sets NCCL_MIN / MAX_NCHANNELS)."""
    from resdepth_amd import _lib
    out = {"bucket_sweep": [], "note": "eager iteration, diagnostics only"}
    keep = gs.bucket_bytes
    for mb in (4, 8, 16):
        gs.bucket_bytes = mb << 20
        gs._model_key = None                              # re-plan the buckets at the next backward
        for _ in range(2):
            tb.step()
        d = diagnostics_pass(tb, gs, steps, barrier)
        out["bucket_sweep"].append({"bucket_mb": mb, "step_ms": d.get("step_ms_under_probe"),
                                    "exposed_grad_allreduce_ms": (d.get("grad_wait") or {}).get("device_ms_per_step"),
                                    "n_buckets": (d.get("plan") or {}).get("n_buckets"),
                                    "bucket_issue_ms_after_step_start": d.get("bucket_issue_ms_after_step_start")})
    gs.bucket_bytes = keep
    gs._model_key = None

    def mfma_ms():
        for _ in range(2):
            tb.step()
        barrier()
        _lib.prof_reset()
        _lib.prof_enable(1)
        for _ in range(steps):
            tb.step()
        torch.cuda.synchronize()
        _lib.prof_enable(0)
        return round(sum(k["ms"] for k in _lib.prof_collect()) / steps, 3)
    with_c = mfma_ms()
    model_gs, tb_gs = tb.model.grad_sync, tb.gs
    tb.model.grad_sync = tb.gs = None                     # the same kernels without any collective (local loss normaliser)
    try:
        without_c = mfma_ms()
    finally:
        tb.model.grad_sync, tb.gs = model_gs, tb_gs
    for _ in range(2):
        tb.step()
    barrier()
    out["mfma_kernel_ms_per_step"] = {"with_collectives_in_flight": with_c, "without_collectives": without_c,
                                      "note": "sum of the MFMA-class kernels' HIP-event durations per step (two-stream backward: "
                                              "overlapping kernels each count their own elapsed time); the difference is what RCCL's "
                                              "co-resident channels cost the convolution kernels"}
    try:
        ver = torch.cuda.nccl.version()
    except Exception:       # noqa: BLE001
        ver = None
    out["rccl"] = {"version": ver, "NCCL_MIN_NCHANNELS": os.environ.get("NCCL_MIN_NCHANNELS"),
                   "NCCL_MAX_NCHANNELS": os.environ.get("NCCL_MAX_NCHANNELS"), "NCCL_ALGO": os.environ.get("NCCL_ALGO"),
                   "NCCL_PROTO": os.environ.get("NCCL_PROTO"), "rccl_channels_flag": args.rccl_channels}
    return out


def summarize_ranks(per_rank_ms, diags, args):
    """rank 0: the per-rank diagnostics -> the `dist` fields (DESIGN.md section 6 says how to read each one)."""
    def col(key, sub=None):
        vals = []
        for d in diags:
            v = (d or {}).get(key)
            if sub is not None and isinstance(v, dict):
                v = v.get(sub)
            vals.append(v)
        return vals

    lo, hi = min(per_rank_ms), max(per_rank_ms)
    out = {"rank_step_ms_min_max_skew": [round(lo, 3), round(hi, 3), round(hi - lo, 3)],
           "slowest_rank": per_rank_ms.index(hi),
           "step_ms_median_hip_events_per_rank": col("step_ms_median_hip_events"),
           "host_enqueue_ms_per_rank": col("host_enqueue_ms"),
           "exposed_grad_allreduce_ms_per_rank": col("grad_wait", "device_ms_per_step"),
           "grad_wait_host_ms_per_rank": col("grad_wait", "host_ms_per_step"),
           "loss_normaliser_allreduce_ms_per_rank": col("loss_norm", "device_ms_per_step"),
           "affinity_per_rank": col("affinity")}
    d0 = diags[0] or {}
    if "tuning" in d0:
        t0 = d0["tuning"]
        out["bucket_sweep_rank0"] = t0.get("bucket_sweep")
        out["rccl"] = t0.get("rccl")
        out["mfma_kernel_ms_per_step_per_rank"] = [((d or {}).get("tuning") or {}).get("mfma_kernel_ms_per_step") for d in diags]
        sweep = [b for b in (t0.get("bucket_sweep") or []) if b.get("step_ms") is not None]
        if sweep:
            best = min(sweep, key=lambda b: b["step_ms"])
            out["bucket_sweep_best"] = {"bucket_mb": best["bucket_mb"], "step_ms": best["step_ms"],
                                        "note": "fastest of the swept bucket sizes on rank 0 (re-run with --bucket-mb to adopt it)"}
    if any((d or {}).get("tuning_error") for d in diags):
        out["tuning_error_per_rank"] = col("tuning_error")
    if "plan" in d0:
        out["gradient_buckets"] = d0["plan"]
    if "bucket_issue_ms_after_step_start" in d0:
        out["bucket_issue_ms_after_step_start_rank0"] = d0["bucket_issue_ms_after_step_start"]
        out["step_ms_under_probe_rank0"] = d0.get("step_ms_under_probe")
    if "bn_fwd" in d0 or "bn_bwd" in d0:
        out["syncbn_rank0"] = {"fwd": d0.get("bn_fwd"), "bwd": d0.get("bn_bwd")}
    out["how_to_read"] = ("exposed_grad_allreduce >> 0.1 ms: buckets finish after the backward (raise --bucket-mb / check "
                          "bucket_issue times against step_ms); loss_normaliser >> 0.05 ms: small-message latency; "
                          "host_enqueue close to step_ms or differing between ranks: host-bound launch thread (affinity); "
                          "skew: a slow rank stalls everyone in the first collective")
    return out


def self_launch(n, argv):
    """`python bench.py --gpus N` without torch.distributed.run: start the N ranks (one process per GPU, LOCAL_RANK = GPU
    index, rendezvous on 127.0.0.1 at a free port) with the same arguments and wait for them.  Rank 0 prints the JSON
    line on the inherited stdout; the return code is 0 only if every rank exited 0 (a failing rank takes the others down,
    by PID, so a dead rank cannot leave the rest waiting in a collective)."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RD_BENCH_SELF_LAUNCHED="1")
        env.setdefault("OMP_NUM_THREADS", "8")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc, alive = 0, set(range(n))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
                for o in alive:
                    procs[o].terminate()
        if alive:
            time.sleep(0.05)
    return rc


def emit_line(record, use_dist):
    """Rank 0's ONE JSON line, as the LAST thing on stdout: the process group is torn down first and the C library's stdout buffer
    is flushed before the line goes out -- RCCL prints a version banner through C stdio when the communicator is created, which on
    a pipe would otherwise surface at exit, after the line."""
    if use_dist:
        import torch.distributed as dist
        dist.destroy_process_group()
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:       # noqa: BLE001
        pass
    sys.stdout.flush()
    if record is not None:
        print(json.dumps(record), flush=True)


def quiet_non_zero_ranks(rank):
    """Ranks other than 0 print nothing on stdout -- neither python nor a C library in their process (RCCL's banner): fd 1 is
    pointed at stderr, whatever launcher started them."""
    if rank != 0:
        sys.stdout.flush()
        os.dup2(2, 1)


def init_dist(args):
    """Process group of the benchmark: RCCL (backend "nccl") -- or, for the --share-gpu code-path check, gloo, whose
    collectives take device tensors in this torch build (staged through the host by the backend itself; nothing from
    tests/ is imported here)."""
    import torch.distributed as dist
    if getattr(args, "rccl_channels", 0) > 0:
        os.environ["NCCL_MIN_NCHANNELS"] = os.environ["NCCL_MAX_NCHANNELS"] = str(args.rccl_channels)
    dist.init_process_group(args.backend or "nccl")


def rendezvous_only(args, world, rank):
    """--rendezvous-only: the launcher path without the workload -- every rank joins the process group (`--backend gloo`
    in the GPU-less build container, nccl = RCCL on the GPU box), all-reduces its rank number and rank 0 prints what it
    saw.  tests/test_host_cpu.py runs `bench.py --gpus 2 --rendezvous-only --backend gloo` with no launcher."""
    import torch.distributed as dist
    backend = args.backend or "nccl"
    if os.environ.get("RD_BENCH_TEST_FAIL_RANK") == str(rank):      # test hook: a rank that dies before the rendezvous
        sys.exit(3)
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend)
    t = torch.tensor([float(rank)], device="cuda" if backend == "nccl" else "cpu")
    dist.all_reduce(t)
    record = None
    if rank == 0:
        record = {"rendezvous": "ok", "backend": dist.get_backend(), "world_size_reported": dist.get_world_size(),
                  "n_gpus": world, "rank_sum": float(t), "self_launched": bool(os.environ.get("RD_BENCH_SELF_LAUNCHED")),
                  "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}
    emit_line(record, True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="tiles per GPU")
    ap.add_argument("--sync-bn", action="store_true", help="SyncBN (single-device-equivalent statistics)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the exact-f32 / cfg-M / cfg-G secondary measurements")
    ap.add_argument("--no-prof", action="store_true", help="disable the per-kernel HIP-event timing")
    ap.add_argument("--graph", action="store_true",
                    help="replay the iteration as one captured HIP graph (resdepth_amd.graph) instead of ~110 eager launches: pays when the "
                         "step is launch-bound (batch <= 6 at cfg-S), costs the two-stream overlap at the benchmark batch")
    ap.add_argument("--no-plan", action="store_true",
                    help="enqueue every iteration from Python instead of replaying the recorded launch plan (resdepth_amd/plan.py)")
    ap.add_argument("--separate-bn-stats", action="store_true",
                    help="A/B: BN-backward sums from the stand-alone reduction pass instead of the data-gradient epilogues")
    ap.add_argument("--serial-backward", action="store_true",
                    help="disable the two-stream backward (weight gradients overlapping the dgrad/BN chain) in the timed region")
    ap.add_argument("--prof-steps", type=int, default=5, help="steps of the serialized, instrumented roofline pass")
    ap.add_argument("--prof-mfma-only", action="store_true",
                    help="roofline pass: bracket only the MFMA kernel classes with HIP events (default: every kernel class)")
    ap.add_argument("--workload", choices=["S", "M"], default="S",
                    help="S = cfg-S (BASELINE configs[1], the headline metric); M = cfg-M (configs[3]: 2-ch 512x512 depth-6)")
    ap.add_argument("--from-rasters", action="store_true",
                    help="draw a fresh augmented batch from HBM-resident synthetic rasters with resdepth_amd.GpuPatchSampler "
                         "inside every timed step (sample assembly + train step) instead of re-using one resident batch")
    ap.add_argument("--infer", action="store_true",
                    help="measure the tiled full-raster inference sweep instead (BASELINE configs[4], cfg-G: 3-ch tiles of "
                         "256x256 at stride 128 over a synthetic --raster x --raster DSM, eval-mode BN, linear blend)")
    ap.add_argument("--raster", type=int, default=8192, help="--infer: side of the synthetic raster (SURVEY 8d: 8192 -> 3969 tiles)")
    ap.add_argument("--bucket-mb", type=int, default=16, help="gradient all-reduce bucket size (resdepth_amd.dp.attach)")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="start the ranks, join the process group, all-reduce one number, print what rank 0 saw and exit")
    ap.add_argument("--backend", default=None, help="process-group backend (default nccl = RCCL; gloo for --rendezvous-only on CPU)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="CODE-PATH CHECK, never a measurement: every rank uses cuda:0 (RCCL refuses two ranks on one device, so "
                         "this needs --backend gloo, which takes device tensors natively in this torch build).  Lets a "
                         "one-GPU box execute `bench.py --gpus N` end to end: launcher, broadcast, bucketed all-reduce, "
                         "max-over-ranks timing, rank-0 line")
    ap.add_argument("--fast-eval", action="store_true",
                    help="--infer: UNet.fast_eval = True, one scale per operand TENSOR as in training (default: one per IMAGE, so that a "
                         "tile's result does not depend on the batch it is in)")
    ap.add_argument("--eval-six", action="store_true",
                    help="--infer: UNet.eval_per_image = False, the six-product split3 bodies of r01-r05")
    ap.add_argument("--no-fold", action="store_true", help="--infer: keep eval-mode BN as separate kernels (A/B of the folded path)")
    ap.add_argument("--prof-all", action="store_true", help="(kept for scripts) same as the default full breakdown")
    ap.add_argument("--rccl-channels", type=int, default=0,
                    help="set NCCL_MIN_NCHANNELS = NCCL_MAX_NCHANNELS = N before the process group is created (0: RCCL's own choice); "
                         "fewer channels leave more CUs / LDS to the convolution kernels the all-reduce overlaps")
    ap.add_argument("--no-tuning", action="store_true", help="skip the bucket-size sweep / CU-contention passes of a multi-rank run")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the data-parallel code path (RCCL process group, bucketed all-reduce) even at world size 1")
    ap.add_argument("--no-pin", action="store_true",
                    help="N > 1: do not bind the rank to the CPUs of its GPU's NUMA node (pin_rank_to_gpu_numa)")
    ap.add_argument("--diag-steps", type=int, default=5,
                    help="steps of the diagnostics pass after the timed region (host enqueue time; N > 1 / --force-dist: exposed "
                         "all-reduce wait, loss-normaliser latency, bucket issue times); 0 = skip")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the driver's plain `python3 bench.py --gpus N ...`: no launcher set the rank environment, so this process becomes
        # the launcher (one rank per GPU over RCCL, exactly what torch.distributed.run --nproc-per-node N would start)
        if not args.rendezvous_only and not args.share_gpu and torch.cuda.device_count() < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but only {torch.cuda.device_count()} HIP device(s) are visible", file=sys.stderr)
            sys.exit(2)
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    if args.gpus != world:
        print(f"bench.py: --gpus {args.gpus} under a launcher that started {world} rank(s) (WORLD_SIZE={world}): "
              f"start it with --nproc-per-node {args.gpus}, or with no launcher at all", file=sys.stderr)
        sys.exit(2)
    quiet_non_zero_ranks(rank)
    if args.rendezvous_only:
        return rendezvous_only(args, world, rank)
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    if args.share_gpu:
        if (args.backend or "nccl") != "gloo":
            print("bench.py: --share-gpu needs --backend gloo (RCCL refuses two ranks on one device)", file=sys.stderr)
            sys.exit(2)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    affinity = None
    if world > 1 and not args.no_pin:
        # before the first kernel launch / OpenMP region: threads started later inherit the mask
        affinity = pin_rank_to_gpu_numa(int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("LOCAL_WORLD_SIZE", world)),
                                        local_rank)

    from resdepth_amd import _lib, dp
    _lib.load()

    if args.infer:
        return infer_main(args, world, rank, dev)
    gs = None
    use_dist = world > 1 or args.force_dist
    dist_info = None
    if use_dist:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:          # --force-dist without a launcher
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        # no device_id=: with the eager communicator initialisation it triggers, every step of this process ran 1.5-2 ms
        # slower on the MI355X boxes (with or without collectives in flight); torch.cuda.set_device above already pins
        # the rank to its GPU for the lazily created communicator
        init_dist(args)
    wl = WORKLOADS[args.workload]
    n = args.batch
    tb = TrainBench(wl, n, dev, rank=rank, from_rasters=args.from_rasters)
    if use_dist:
        gs = dp.attach(tb.model, sync_bn=args.sync_bn, bucket_bytes=args.bucket_mb << 20)
        tb.gs = gs
        dp.broadcast_parameters(tb.model, 0)
    tb.attach_optimizer()
    tb.model.two_stream_backward = not args.serial_backward
    tb.model.fused_bn_bwd_stats = not args.separate_bn_stats

    def barrier():
        if use_dist:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    tb.use_graph = args.graph and not use_dist
    tb.use_plan = not args.no_plan and not tb.use_graph
    if tb.use_graph or tb.use_plan:
        for _ in range(5):                       # two eager warm-up iterations, capture preparation, capture (+ the plan's verified
            tb.step()                            # first replay), one plain replay: outside the timed region
        torch.cuda.synchronize()
    dt, step_ms = tb.timed(args.steps, args.warmup, barrier)
    timed_losses = list(tb.losses)
    plan_info = {"enabled": bool(tb.use_plan), "replays": tb.plan.replays, "launches": tb.plan.n_launches, "segments": tb.plan.n_segments,
                 "last_eager_reason": tb.plan.why_eager, "rejected": getattr(tb.plan, "plan_rejected", None),
                 "note": "the iteration's ~110 kernel launches recorded once (under a graph capture, for its reserved memory) and "
                         "replayed from C with one hipLaunchKernel each onto the REAL two streams (resdepth_amd/plan.py, rd_plan_*): "
                         "the eager iteration's kernels, arguments, stream overlap and bits (the first replay is verified bit for bit "
                         "against an eager iteration; tests/test_plan_gpu.py), without its Python / ctypes enqueue time; under data "
                         "parallelism the collectives are issued by the host between the plan's segments"}
    graph_info = {"enabled": bool(tb.use_graph), "replays": tb.graph.replays, "last_eager_reason": tb.graph.why_eager,
                  "note": "the whole iteration (weight packing, forward, loss, two-stream backward, Adam) replayed as one captured HIP "
                          "graph per step: the same kernels and arguments as the eager iteration, bit-identical results "
                          "(tests/test_graph_gpu.py).  Off by default: at this batch the eager iteration is not launch-bound and its "
                          "two-stream backward overlaps better than the graph's branches (secondary.hip_graph_small_batch has the "
                          "launch-bound case)"}
    # ---- diagnostics pass (after the timed region, production two-stream mode): host enqueue time and, with a process
    # group, the exposed part of every exchange point
    diag = diagnostics_pass(tb, gs, args.diag_steps, barrier) if args.diag_steps > 0 else None
    eager_info = None
    if (tb.use_graph or tb.use_plan) and diag is not None:        # the eager iteration beside the replayed one: enqueue time and rate
        tb.use_graph = tb.use_plan = False
        d2 = diagnostics_pass(tb, gs, args.diag_steps, barrier)
        diag["host_enqueue_ms_eager"] = d2["host_enqueue_ms"]
        dte, eve = tb.timed(min(10, args.steps), 2, barrier)
        eager_info = {"tiles_per_s": round(n * world * min(10, args.steps) / dte, 1), "step_ms_median": round(_median(eve), 3),
                      "host_enqueue_ms": d2["host_enqueue_ms"],
                      "note": "the same iteration enqueued from Python (the r01-r05 headline path), measured right after the timed region"}
    tb.use_graph = tb.use_plan = False           # everything below (instrumented passes, arithmetic switches) runs eagerly
    if use_dist and gs is not None and diag is not None and not args.no_tuning:
        try:
            diag["tuning"] = dist_tuning_pass(tb, gs, args, barrier, max(2, min(args.diag_steps, 4)))
        except Exception as e:      # noqa: BLE001 -- diagnostics of the first multi-GPU run must never cost it its headline line
            # (every rank runs the same code on the same shapes, so a failure here is the same failure everywhere)
            diag["tuning_error"] = f"{type(e).__name__}: {str(e)[:300]}"
            tb.model.grad_sync, tb.gs = gs, gs
            gs.probe = None
    # ---- per-kernel roofline pass.  In the timed region the weight-gradient kernels run concurrently with the
    # dgrad/BN chain on a second stream, so a kernel's event-to-event duration there includes time shared with another
    # kernel; the roofline numbers therefore come from a SERIALIZED pass of the same step, in this process, right after
    # the timed region (HIP events on the launch stream, rd_prof_*).  `value` is never taken from this pass.
    kern, prof_steps = [], 0
    if not args.no_prof:
        tb.model.two_stream_backward = False
        tb.step()
        torch.cuda.synchronize()
        _lib.prof_reset()
        _lib.prof_enable(1 if args.prof_mfma_only else 2)
        prof_steps = max(1, args.prof_steps)
        for _ in range(prof_steps):
            tb.step()
        torch.cuda.synchronize()
        _lib.prof_enable(0)
        kern = _lib.prof_collect()
        tb.model.two_stream_backward = not args.serial_backward
    loss_vals = [float(v) for v in timed_losses]
    per_rank_ms = [dt / args.steps * 1e3]
    if use_dist:
        import torch.distributed as dist
        mine = torch.tensor([dt], device=dev, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(every, mine)
        per_rank_ms = [float(v) / args.steps * 1e3 for v in every]
        dt = max(float(v) for v in every)               # MAX over ranks
        mine_diag = dict(diag or {}, step_ms_median_hip_events=round(_median(step_ms), 3), affinity=affinity, rank=rank)
        every_diag = [None] * dist.get_world_size()
        dist.all_gather_object(every_diag, mine_diag)
        dist_info = {"backend": dist.get_backend(), "world_size_reported": dist.get_world_size(),
                     "per_rank_ms_per_step": [round(v, 3) for v in per_rank_ms]}
        dist_info.update(summarize_ranks(per_rank_ms, every_diag, args))

    if rank == 0:
        ms = dt / args.steps * 1e3
        tiles_s = n * world * args.steps / dt
        kernels = kernel_rows(kern, prof_steps)
        roof = build_roofline(kern, prof_steps, PMC_SUMMARIES[args.workload],
                              f"HIP events, serialized pass of {prof_steps} steps right after the timed region "
                              "(timed region itself: un-instrumented, wgrad kernels overlapped on a 2nd stream)")
        per_gpu = tiles_s / world
        from resdepth_amd import _lib as _l
        mode = _l.mfma_mode()
        out = {
            "metric": "DSM tiles/sec fwd+bwd (256x256, 3-ch, depth-5 U-Net)" if args.workload == "S" else
                      "DSM tiles/sec fwd+bwd (512x512, 2-ch, depth-6 U-Net)", "value": round(tiles_s, 2),
            "unit": "tiles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "arithmetic_mode": mode,
            "arithmetic": ARITHMETIC_SPLIT2H if mode == "split2h" else ARITHMETIC_SPLIT3 if mode == "split3" else "RD_MFMA=f32: v_mfma_f32_32x32x2_f32, exact",
            "data": ("synthetic rasters resident in HBM, a fresh augmented batch assembled on the GPU every step"
                     if args.from_rasters else "synthetic (randn tiles resident in HBM, default-initialised weights)"),
            "config": {"workload": wl["name"] + ", fwd+loss+bwd+Adam",
                       "tiles_per_gpu": n, "global_batch": n * world,
                       "parallelism": f"dp{world}" + ("+syncbn" if args.sync_bn else "") +
                                      (" (ranks SHARE one GPU over gloo: launcher / data-parallel code-path check, not a scaling "
                                       "number)" if args.share_gpu else ""),
                       "backward": "serial" if args.serial_backward else "two-stream (wgrad || dgrad+BN)"},
            "step_ms_median": round(_median(step_ms), 3),
            "step_ms_min_max": [round(min(step_ms), 3), round(max(step_ms), 3)],
            "step_timing": "ms_per_step/value: wall clock over the K steps between barriers (max over ranks); step_ms_*: one HIP-event "
                           "pair per step on the launch stream (rank 0)",
            "e2e": {"tflops": round(per_gpu * wl["flop"] / 1e12, 2),
                    "frac_f32_peak": round(per_gpu * wl["flop"] / 1e12 / PEAK_F32_TFLOPS, 4),
                    "frac_split_mfma_bound": round(per_gpu * wl["flop"] / 1e12 / peak_split_tflops(3 if mode == "split2h" else 6), 4),
                    "hbm_frac": round(per_gpu * wl["bytes_a"] / (PEAK_HBM_GBS * 1e9), 4),
                    "hbm_frac_note": "tiles/s/GPU x op-level compulsory bytes per tile (SURVEY 8d model A) / 8 TB/s"},
            "roofline": roof,
            "kernels": kernels,
            "loss_first_last": [round(loss_vals[0], 6), round(loss_vals[-1], 6)] if loss_vals else None,
        }
        out["hip_graph"] = graph_info
        out["launch_plan"] = plan_info
        if eager_info is not None:
            out["eager_iteration"] = eager_info
        if dist_info:
            out["dist"] = dist_info
        if diag is not None:
            out["host_enqueue_ms"] = diag["host_enqueue_ms"]
            if "host_enqueue_ms_eager" in diag:
                out["host_enqueue_ms_eager"] = diag["host_enqueue_ms_eager"]
            out["host_enqueue_note"] = ("wall time of one step() on the launching thread (median of the diagnostics pass, GPU "
                                        "queue never full): every kernel launch, allocation and event of the step through ctypes")
        if world == 1 and not args.no_secondary and args.workload == "S" and not args.from_rasters and not use_dist:
            out["secondary"] = secondary_measurements(args, dev, tb)
        if world == 1 and not args.no_cpu_baseline and args.workload == "S":
            out["cpu_baseline"] = cpu_baseline()
    emit_line(out if rank == 0 else None, use_dist)


if __name__ == "__main__":
    main()
