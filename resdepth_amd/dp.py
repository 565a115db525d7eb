"""Data parallelism for the ResDepth train step: one process per GPU, RCCL over xGMI through
torch.distributed (backend "nccl" is RCCL on ROCm; "gloo" on CPU tensors for the no-GPU tests).

The reference has no distributed code at all (single `cuda:0`, lib/Trainer.py:34); this is new
functionality required by the north star.  Tiles are independent, so the batch is sharded and the
path has exactly three exchange points:

  1. gradients   -- SUM all-reduce of the flat fp32 gradient buffer (12.6 M floats = 50.5 MB for
                    cfg-S), cut into a few buckets that are launched asynchronously as soon as the
                    backward pass has produced them (the backward fills the flat buffer from its END
                    towards its start), so the transfer overlaps the remaining wgrad/dgrad GEMMs;
  2. loss        -- the reference divides by the number of valid pixels of the WHOLE batch
                    (lib/Trainer.py:98), so (sum|d|, sum mask) are all-reduced BEFORE the backward and
                    every rank scales its gradient by 1/global-count: local gradients then SUM to the
                    single-device gradient (no averaging);
  3. BatchNorm   -- optional SyncBN: per layer all-reduce of (sum x, sum x^2) in the forward and of
                    (sum g, sum g*xhat) in the backward (2*C doubles each), which makes N-GPU training
                    numerically equivalent to the reference's single-device batch.  Off by default
                    (plain DDP semantics, local statistics).

xGMI is point-to-point (7 links x ~153 GB/s per GPU): a 50 MB ring all-reduce is ~0.6 ms per-link
bound -- small against a >=20 ms step -- so a handful of >=8 MB buckets is the right granularity;
finer buckets only add launch latency.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist

from . import _lib


class Probe:
    """Optional per-step timing of the exchange points, for the benchmark's `dist` block (never on by default: it
    records two HIP events per bracket on the stream the collective is ordered on).  A bracket is the time THAT stream
    spends between "everything before the collective is done" and "the collective's result is visible to it", i.e. the
    part of the collective the stream could not hide:
      grad_wait   GradSync.finish: main stream blocked on the gradient buckets after its own backward work (and the
                  weight-gradient stream's) has finished -- the EXPOSED part of the all-reduce;
      loss_norm   the two-scalar loss-normaliser all-reduce (blocking, between the loss reduction and its finish kernel);
      bn_fwd / bn_bwd   SyncBN statistics exchanges (2 x 10 per step in strict-parity mode).
    Host values: `host_ms[name]` = wall time the launching thread spent inside the bracket (enqueue + any host wait)."""

    def __init__(self):
        self.pairs = {}
        self.host_ms = {}
        self.launch_ev = []          # (bucket index, event recorded where the bucket's collective was issued)
        self.step_start = None

    class _Bracket:
        def __init__(self, probe, name):
            self.probe, self.name = probe, name

        def __enter__(self):
            import time
            self.a = torch.cuda.Event(enable_timing=True)
            self.a.record()
            self.t0 = time.perf_counter()

        def __exit__(self, *exc):
            import time
            b = torch.cuda.Event(enable_timing=True)
            b.record()
            self.probe.pairs.setdefault(self.name, []).append((self.a, b))
            self.probe.host_ms.setdefault(self.name, []).append((time.perf_counter() - self.t0) * 1e3)

    def bracket(self, name):
        return Probe._Bracket(self, name)

    def mark_step_start(self):
        self.step_start = torch.cuda.Event(enable_timing=True)
        self.step_start.record()

    def mark_bucket(self, bi, stream):
        if self.step_start is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(stream)
            self.launch_ev.append((bi, self.step_start, ev))

    def summary(self, steps: int) -> dict:
        """Call after torch.cuda.synchronize().  Per-step means in ms."""
        out = {}
        for name, pairs in self.pairs.items():
            ms = [a.elapsed_time(b) for a, b in pairs]
            out[name] = {"calls_per_step": round(len(ms) / max(1, steps), 2), "device_ms_per_step": round(sum(ms) / max(1, steps), 4),
                         "device_ms_max_call": round(max(ms), 4),
                         "host_ms_per_step": round(sum(self.host_ms[name]) / max(1, steps), 4)}
        if self.launch_ev:
            by = {}
            for bi, s, e in self.launch_ev:
                by.setdefault(bi, []).append(s.elapsed_time(e))
            out["bucket_issue_ms_after_step_start"] = {str(bi): round(sum(v) / len(v), 3) for bi, v in sorted(by.items())}
        return out


class _NoBracket:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_BRACKET = _NoBracket()


class GradSync:
    def __init__(self, process_group=None, bucket_bytes: int = 16 << 20, tail_bytes: int = 512 << 10):
        if not dist.is_available() or not dist.is_initialized():
            raise RuntimeError("GradSync needs an initialised torch.distributed process group")
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self.bucket_bytes = int(bucket_bytes)
        self.tail_bytes = int(tail_bytes)      # size cap of the final (exposed) bucket; 0 = the leftover as it comes
        self._buckets = None      # list of dicts: lo, hi (element offsets), params (set of indices)
        self._ready = set()
        self._launched = []
        self._handles = []
        self._model_key = None
        self.launch_stream = None   # set by the two-stream backward: the stream the wgrad kernels run on
        self.probe: Optional[Probe] = None      # bench.py's diagnostics pass sets a Probe; None = no events recorded

    def _t(self, name):
        return self.probe.bracket(name) if self.probe is not None else _NO_BRACKET

    def describe(self) -> dict:
        """Static facts of the exchange plan (after the first backward): bucket count / sizes in the order they are issued."""
        b = self._buckets or []
        return {"world": self.world, "bucket_bytes_target": self.bucket_bytes, "n_buckets": len(b),
                "bucket_mbytes": [round((x["hi"] - x["lo"]) * 4 / 2 ** 20, 2) for x in b],
                "bucket_params": [len(x["params"]) for x in b]}

    # ---- small collectives ---------------------------------------------------------------------
    def allreduce_loss_sums(self, sums: torch.Tensor, numel: int) -> int:
        """(sum |d|, #valid) -> global; returns the global element count."""
        def run():
            with self._t("loss_norm"):
                dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=self.pg)
        _lib.host_action(run)            # (a launch plan re-issues it between its segments: resdepth_amd/plan.py)
        return numel * self.world

    def allreduce_stats(self, sums: torch.Tensor, count: int) -> int:
        """SyncBN forward: (sum x, sum x^2) per channel -> global; returns the global pixel count.  Every rank must
        hold the same per-rank batch: `shard_batch` guarantees it, `check_equal_across_ranks` (called by the Trainer
        for its loaders) verifies it for user-built shards -- a smaller batch on one rank would silently bias the
        statistics, since the count is not part of the exchange (it stays a host value; no device read-back)."""
        with self._t("bn_fwd"):
            dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=self.pg)
        return count * self.world

    def check_equal_across_ranks(self, value: int, what: str) -> None:
        """Raise on every rank if `value` differs between ranks (loader lengths, batch sizes): unequal shards would
        deadlock the per-step collectives or bias the SyncBN statistics."""
        dev = "cuda" if dist.get_backend(self.pg) == "nccl" else "cpu"
        mine = torch.tensor([int(value)], dtype=torch.int64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(every, mine, group=self.pg)
        vals = [int(v) for v in every]
        if any(v != vals[0] for v in vals):
            raise RuntimeError(f"resdepth_amd.dp: {what} differs across ranks ({vals}); every rank needs the same number of "
                               "equally sized batches (use drop_last=True / equal shards)")

    def allreduce_sums(self, sums: torch.Tensor) -> None:
        """SyncBN backward: (sum g', sum g'*xhat, ...) -> global, in place."""
        with self._t("bn_bwd"):
            dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=self.pg)

    # ---- gradient buckets ------------------------------------------------------------------------
    @staticmethod
    def plan_buckets(offsets: List[int], sizes: List[int], bucket_elems: int, tail_elems: int = 0):
        """Contiguous flat ranges covering every parameter, built from the END of the buffer (the
        order the backward produces gradients in).  Pure function (unit-tested on CPU).

        tail_elems > 0: the LAST bucket -- the one whose all-reduce nothing can hide, it is issued when the backward is
        over -- is cut again, geometrically: its final piece holds at most `tail_elems` elements (whole tensors; at least
        one), the piece before it at most 4 x that, and so on.  cfg-S with 512 KB: the leftover [enc3 .. enc0] = 5.9 MB becomes
        [enc3] 4.5 MB (issued ~1.8 ms before the end of the step), [enc2] 1.2 MB (~0.5 ms before), [enc1, enc0] 0.3 MB exposed."""
        order = sorted(range(len(offsets)), key=lambda i: offsets[i], reverse=True)
        buckets = []
        cur = None
        for i in order:
            lo, hi = offsets[i], offsets[i] + sizes[i]
            if cur is None:
                cur = {"lo": lo, "hi": hi, "params": {i}}
            else:
                assert hi == cur["lo"], "parameters must tile the flat buffer"
                cur["lo"] = lo
                cur["params"].add(i)
            if cur["hi"] - cur["lo"] >= bucket_elems:
                buckets.append(cur)
                cur = None
        if cur is not None:
            buckets.append(cur)
        # geometric tail: the final piece <= tail_elems, the one before it <= 4 x that, ... until what is left fits its cap
        cap, pieces = int(tail_elems), []
        while cap > 0 and buckets and buckets[-1]["hi"] - buckets[-1]["lo"] > cap and len(buckets[-1]["params"]) > 1:
            last = buckets.pop()
            members = sorted(last["params"], key=lambda i: offsets[i])          # from the START of the buffer = completed last
            tail, n = set(), 0
            for i in members:
                if tail and n + sizes[i] > cap:
                    break
                tail.add(i)
                n += sizes[i]
            if len(tail) == len(members):
                buckets.append(last)
                break
            cut = last["lo"] + n
            pieces.append({"lo": last["lo"], "hi": cut, "params": tail})
            buckets.append({"lo": cut, "hi": last["hi"], "params": last["params"] - tail})
            cap *= 4
        buckets.extend(reversed(pieces))
        return buckets

    def _ensure_plan(self, model):
        key = (id(model), model._flat_grad.data_ptr(), model._flat_grad.numel())
        if self._model_key != key:
            sizes = [p.numel() for p in model.parameters()]
            self._buckets = self.plan_buckets(list(model._offsets), sizes, max(1, self.bucket_bytes // 4), self.tail_bytes // 4)
            self._model_key = key
            self._ready = set()
            self._launched = [False] * len(self._buckets)
            self._handles = []

    def _launch(self, model, bi):
        b = self._buckets[bi]
        view = model._flat_grad[b["lo"]:b["hi"]]
        launch_stream = self.launch_stream

        def issue():
            if launch_stream is not None and view.is_cuda:
                # a bucket mixes gradients produced on the main stream (BN / bias) and on the wgrad stream: the collective
                # is issued from the wgrad stream after it has caught up with the main stream (cheap: main runs ahead)
                cur = torch.cuda.current_stream()
                with torch.cuda.stream(launch_stream):
                    if cur != launch_stream:
                        launch_stream.wait_stream(cur)
                    h = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
            else:
                h = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
            self._handles.append(h)
            if self.probe is not None and view.is_cuda:
                self.probe.mark_bucket(bi, launch_stream if launch_stream is not None else torch.cuda.current_stream())
        # eager: issued now; while a launch plan is recorded: the plan's current segment ends here and the replay loop issues
        # the collective at this point of every iteration (the view is a range of the persistent flat gradient buffer)
        _lib.host_action(issue)
        self._launched[bi] = True

    def params_ready(self, model, indices) -> None:
        """Called by the backward pass after the kernels writing these parameter gradients were enqueued."""
        self._ensure_plan(model)
        self._ready.update(indices)
        # every rank runs the same backward code, so buckets complete -- and their collectives are issued --
        # in the same order everywhere (the bucket holding the last ConvTranspose2d bias completes last)
        for bi, b in enumerate(self._buckets):
            if not self._launched[bi] and b["params"] <= self._ready:
                self._launch(model, bi)

    def finish(self, model) -> None:
        """Launch whatever is left, then make the current stream wait for every bucket."""
        self._ensure_plan(model)
        for bi in range(len(self._buckets)):
            if not self._launched[bi]:
                self._launch(model, bi)
        def wait_all():
            with self._t("grad_wait"):
                for h in self._handles:
                    h.wait()
            self._handles = []
        _lib.host_action(wait_all)
        self._ready = set()
        self._launched = [False] * len(self._buckets)

    def allreduce_unbucketed(self, flat: torch.Tensor) -> None:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.pg)


def attach(model, process_group=None, sync_bn: bool = False, bucket_bytes: int = 16 << 20, tail_bytes: int = 512 << 10) -> GradSync:
    """Make `model` (a resdepth_amd.UNet) data parallel: its backward all-reduces gradients, its loss must be
    built with `grad_sync=` the returned object (global normaliser)."""
    gs = GradSync(process_group, bucket_bytes, tail_bytes)
    model.grad_sync = gs
    model.sync_bn = bool(sync_bn)
    return gs


def broadcast_parameters(model, src: int = 0, process_group=None) -> None:
    """Rank `src`'s parameters and BN buffers to everyone (start of training / after loading a checkpoint).  The
    parameters go as ONE message when the model keeps them in its flat buffer; the packed GEMM-layout weight copies of
    every rank are invalidated (the broadcast writes through `.data`, which autograd's version counters do not see)."""
    flat = getattr(model, "_flat_param", None)
    params = list(model.parameters())
    lo, hi = (flat.data_ptr(), flat.data_ptr() + flat.numel() * 4) if flat is not None else (0, 0)
    if flat is not None and params and all(lo <= t.data_ptr() < hi for t in params) \
            and sum(t.numel() for t in params) == flat.numel():
        dist.broadcast(flat, src=src, group=process_group)       # every parameter is a view of the flat buffer (any layout)
    else:
        for t in params:
            dist.broadcast(t.data, src=src, group=process_group)
    for t in model.buffers():
        dist.broadcast(t.data, src=src, group=process_group)
    _lib.bump_param_generation(None)
    # ... and the generation of every tensor written: a graph whose forward ran BEFORE the broadcast must not run its backward on
    # the new weights (UNet._engine_backward compares the model's own parameter key, which reads these)
    for t in params:
        _lib.bump_param_generation(t.data_ptr())
    if hasattr(model, "invalidate_packed"):
        model.invalidate_packed()


def shard_batch(batch: dict, rank: int, world: int) -> dict:
    """Contiguous shard of a DataLoader-collated batch dict (lib/DsmOrthoDataset.py:281-291)."""
    n = batch["input"].shape[0]
    if n % world != 0:
        raise ValueError(f"global batch {n} is not divisible by the world size {world}")
    per = n // world
    sl = slice(rank * per, (rank + 1) * per)
    return {k: (v[sl] if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == n else v) for k, v in batch.items()}
