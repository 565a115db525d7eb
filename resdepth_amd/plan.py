"""The training iteration replayed from a recorded LAUNCH PLAN (r06).

`GraphedTrainStep` (resdepth_amd/graph.py) replays the iteration as one hipGraph: 0.35 ms of host time, but the graph's
parallel branches do not co-run like two HIP streams do (batch 32: -4.5 % against the eager two-stream iteration) and a graph
cannot hold the bucketed RCCL all-reduce that data parallelism issues from inside the backward -- so the eager iteration, with
3-6 ms of Python + ctypes per step, stayed the default.  A launch plan keeps the REAL streams:

  * record (once): the iteration runs under a graph capture -- nothing executes, torch's allocator serves it from the capture's
    private pool, which stays reserved afterwards and whose reuse pattern is safe across the two streams -- while the library
    notes every kernel launch on the two streams (function, grid, block, argument values), every event record / wait between
    them (`_lib.Ev`, `_lib.ev_wait`), and where the host acts (`_lib.host_action`: the loss normaliser's all-reduce, each
    gradient bucket's all-reduce, the final wait);
  * replay: `rd_plan_replay(plan, segment, main, side)` enqueues a segment's launches from C, one hipLaunchKernel each, onto
    torch's current stream and the model's weight-gradient stream -- the same two-stream overlap as the eager iteration -- and
    between segments the host actions run as in the eager iteration (same torch.distributed calls on the same tensors).

The captured hipGraph itself is never launched; it only owns the memory.  Everything that varies per step lives in device
memory (graph.py: the static batch tensors, Adam's scalar block).  The FIRST replay is verified: the same batch runs once
eagerly and once through the plan from the same state, and parameters, moments, BatchNorm buffers and loss must agree bit for
bit -- a kernel the recorder cannot see (a torch op inside the iteration) would show up there, and the plan is then dropped for
good (`why_eager`).

lib/Trainer.py:159-179,212-222 is the loop this replaces."""
from __future__ import annotations

import ctypes

import torch

from . import _lib
from .graph import GraphedTrainStep


class _Recorder:
    def __init__(self):
        self.actions = []

    def add_action(self, fn):
        seg = _lib.load().rd_plan_segment()
        assert seg == len(self.actions), (seg, len(self.actions))
        self.actions.append(fn)


class PlannedTrainStep(GraphedTrainStep):
    def __init__(self, model, optimizer, warmup: int = 2, keep_grads: bool = False, verify: bool = True):
        super().__init__(model, optimizer, warmup, keep_grads)
        self._plan = None
        self._actions = None
        self._rec = None
        self.verify = verify
        self._verified = False
        self.n_launches = self.n_segments = 0

    # ---- eligibility: data parallelism is fine (host actions); what runs torch ops inside the iteration is not ----------------
    def _eligible(self, x):
        m = self.model
        if not x.is_cuda:
            return "batch not on a HIP device"
        if not m.training:
            return "model in eval mode"
        from .optim import FusedAdam
        if not isinstance(self.optimizer, FusedAdam):
            return f"{type(self.optimizer).__name__} has no captured form"
        if getattr(m, "sync_bn", False) and getattr(m, "grad_sync", None) is not None:
            return "SyncBN exchanges its statistics through torch ops inside the iteration"
        if m._needs_twin() or m._first_generic():
            return "zero-padded twin / generic first convolution copy through torch ops"
        if m.do_outer_skip and m.do_outer_skip_BN:
            return "outer-skip BatchNorm2d(1) is evaluated with torch ops"
        if "prelu" in (m.act_fn_encoder, m.act_fn_decoder, m.act_fn_bottleneck):
            return "the PReLU slope gradient is written with a torch copy"
        if m.up_mode != "transpose" or not m.do_BN:
            return "bilinear up-mode / do_BN=False run per-layer pack launches with a device-to-device copy"
        if _lib.tune_get("mfma_f32"):
            return "exact-f32 mode packs layer by layer"
        if _lib.prof_level_py() != 0:
            return "the per-kernel profiler is on"
        return None

    def invalidate(self):
        super().invalidate()
        if self._plan is not None:
            _lib.load().rd_plan_free(self._plan)
        self._plan = self._actions = None
        self._verified = False

    def __del__(self):
        try:
            if self._plan is not None:
                _lib.load().rd_plan_free(self._plan)
        except Exception:      # noqa: BLE001
            pass
        super().__del__()

    # ---- recording ------------------------------------------------------------------------------------------------------------
    def _side(self):
        m = self.model
        dev = next(m.parameters()).device
        if m._side_stream is None or m._side_stream.device != dev:
            m._side_stream = torch.cuda.Stream(device=dev)
        return m._side_stream

    def _record_begin(self):
        lib = _lib.load()
        side = self._side()
        _lib.check(lib.rd_plan_begin(self._stream.cuda_stream, side.cuda_stream), "plan_begin")
        self._rec = _Recorder()
        _lib._plan_rec = self._rec

    def _record_end(self):
        lib = _lib.load()
        _lib._plan_rec = None
        nl, ns = ctypes.c_int(0), ctypes.c_int(0)
        plan = lib.rd_plan_end(ctypes.cast(ctypes.byref(nl), ctypes.c_void_p), ctypes.cast(ctypes.byref(ns), ctypes.c_void_p))
        if self._plan is not None:
            lib.rd_plan_free(self._plan)
        self._plan = plan
        self._actions = self._rec.actions if plan else None
        self._rec = None
        self.n_launches, self.n_segments = nl.value, ns.value
        if not plan:
            msg = lib.rd_last_error_string()
            self._plan_error = msg.decode() if msg else "recording failed"

    def _capture(self, batch):
        self._plan_error = None
        try:
            super()._capture(batch)
        except _PlanUnavailable:
            raise
        except Exception as e:      # noqa: BLE001 -- whatever a capture can throw (a torch op that refuses to be captured, an
            # allocator or runtime error): the recording executed nothing, so the eager iteration can take this call
            _lib._plan_rec = None
            self.invalidate()
            self.warmup = 1 << 62
            raise _PlanUnavailable(f"the recording raised {type(e).__name__}: {str(e)[:200]}") from e
        if self._plan is None:
            # the iteration holds something a plan cannot: stay eager from now on (the hipGraph of the capture is dropped too)
            why = self._plan_error
            self.invalidate()
            self.warmup = 1 << 62
            raise _PlanUnavailable(why)

    # ---- replay ---------------------------------------------------------------------------------------------------------------
    def _run_plan(self):
        lib = _lib.load()
        main = torch.cuda.current_stream()
        side = self._side()
        m, s = main.cuda_stream, side.cuda_stream
        acts = self._actions
        for seg in range(self.n_segments):
            rc = lib.rd_plan_replay(self._plan, seg, m, s)
            if rc:
                _lib.check(rc, "plan_replay")
            if seg < len(acts):
                acts[seg]()
        self.model._bn_gen += 1

    def _snapshot(self):
        m, o = self.model, self.optimizer
        st = [m._flat_param.clone()] + [b.clone() for b in m.buffers()]
        for gi in sorted(o._flat_state):
            st += [o._flat_state[gi][1].clone(), o._flat_state[gi][2].clone()]
        return st

    def _restore(self, st):
        m, o = self.model, self.optimizer
        it = iter(st)
        m._flat_param.copy_(next(it))
        for b in m.buffers():
            b.copy_(next(it))
        for gi in sorted(o._flat_state):
            o._flat_state[gi][1].copy_(next(it))
            o._flat_state[gi][2].copy_(next(it))

    def _replay(self):
        if self._verified or not self.verify:
            self._run_plan()
            return
        # first replay: the plan against the eager iteration on the same batch from the same state, bit for bit.  (The scalar
        # block of Adam for THIS step was written by optimizer.advance() already; the eager reference takes the same numbers by
        # running rd_adam_step_dev through capture_step.)
        before = self._snapshot()
        self._run_plan()
        torch.cuda.synchronize()
        got = self._snapshot() + [self._loss.clone()]
        self._restore(before)
        for p in self.params:
            p.grad = None
        self.model.invalidate_packed()
        out = self.model(self._static[0])
        from .loss import masked_l1_loss
        loss = masked_l1_loss(out, self._static[1], self._static[2], self._static[3], self._static[4],
                              grad_sync=getattr(self.model, "grad_sync", None))
        loss.backward()
        self.optimizer.capture_step()
        torch.cuda.synchronize()
        want = self._snapshot() + [loss.detach().reshape(())]
        ok = all(torch.equal(a, b) for a, b in zip(got, want))
        gs = getattr(self.model, "grad_sync", None)
        if gs is not None:                       # every rank must take the same decision
            import torch.distributed as dist
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self._static[0].device if dist.get_backend(gs.pg) == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=gs.pg)
            ok = bool(int(flag))
        for p in self.params:
            p.grad = None
        self._loss.copy_(want[-1])               # the state now is the eager iteration's (identical when ok)
        self.model.invalidate_packed()
        if not ok:
            self.invalidate()
            self.warmup = 1 << 62
            self.plan_rejected = "the first replay did not reproduce the eager iteration bit for bit"
            return
        self._verified = True

    def __call__(self, x, y, mask, mean, std):
        try:
            out = super().__call__(x, y, mask, mean, std)
        except _PlanUnavailable as e:
            self.why_eager = f"plan unavailable: {e}"
            self.plan_rejected = str(e)
            return self._eager(x, y, mask, mean, std)
        if getattr(self, "plan_rejected", None) and self.why_eager is None:
            self.why_eager = self.plan_rejected
        return out


class _PlanUnavailable(RuntimeError):
    pass
