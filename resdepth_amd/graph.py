"""One training iteration as ONE HIP graph launch.

The reference's iteration (lib/Trainer.py:212-222: forward, masked L1, backward, optimizer step, grad = None) is ~110
kernel launches here, enqueued through ctypes in 3.3-4.0 ms of host time -- below the 10 ms the GPU needs at batch 32, so the
step is not launch-bound there, but it IS at small batches (batch 4: 3.5 ms of enqueueing for 2.6 ms of kernels) and the
launching thread is the one resource eight ranks on one host contend for.  `GraphedTrainStep` captures the whole iteration --
weight (re)packing, forward, loss, the two-stream backward, FusedAdam -- into a `torch.cuda.CUDAGraph` (hipGraph) once and
replays it: 0.3 ms of host time per step, the same kernels with the same arguments, bit-identical results
(tests/test_graph_gpu.py).

What varies from step to step lives in device memory the graph reads, never in the graph:
  * the batch: five static tensors the caller's batch is copied into (device to device, 36 MB at cfg-S batch 32: ~15 us);
  * Adam's host-side scalars (bias corrections from the step count, a scheduler's learning rate, grad_scale): an 8-float
    block rewritten before every replay (FusedAdam.advance -> rd_adam_step_dev);
  * BatchNorm's running statistics / num_batches_tracked, the parameters, the moments: device state all along.

Not captured -- these calls run the eager iteration instead, with the same results: the first `warmup` calls (they size every
lazily allocated buffer: packed weights, split-K scratch, moments), a batch of another shape (the ragged last batch of an
epoch), a model in eval mode, gradient synchronisation across ranks (the bucketed RCCL all-reduce is issued from autograd
hooks and stays eager), optimizers other than FusedAdam on its flat path.
"""
from __future__ import annotations

import torch

from . import _lib
from .loss import masked_l1_loss
from .optim import FusedAdam


class GraphedTrainStep:
    def __init__(self, model, optimizer, warmup: int = 2, keep_grads: bool = False):
        self.model, self.optimizer = model, optimizer
        self.warmup = max(1, int(warmup))        # >= 1: the moments and the flat gradient buffer exist after one eager step
        self.keep_grads = keep_grads             # leave p.grad pointing at the graph's gradient buffers (default: None, as the reference)
        self.params = [p for g in optimizer.param_groups for p in g["params"]]
        self._calls = 0
        self._graph = None
        self._key = None
        self._static = None
        self._loss = None
        self._grads = None
        self._cap_token = None
        self._stream = None                      # the capture stream, with a split-K scratch of its own (_lib.pin_splitk_workspace)
        self._scratch = None
        self._one = None
        self.why_eager = None                    # reason the last call ran eagerly (None: it was a replay)
        self.replays = 0

    # ---- the eager iteration (lib/Trainer.py:212-222) -------------------------------------------------------------------
    def _eager(self, x, y, mask, mean, std):
        # gradients a previous call left in place (keep_grads, or a replay's graph-owned buffers) must not be ACCUMULATED into:
        # every iteration starts from "no gradient", like lib/Trainer.py:221-222; keep_grads only decides what is left afterwards
        for p in self.params:
            p.grad = None
        out = self.model(x)
        loss = masked_l1_loss(out, y, mask, mean, std, grad_sync=getattr(self.model, "grad_sync", None))
        loss.backward()
        self.optimizer.step()
        if not self.keep_grads:
            for p in self.params:
                p.grad = None
        return loss.detach()

    # ---- hooks of the launch-plan variant (resdepth_amd/plan.py) -----------------------------------------------------------
    def _record_begin(self):
        pass

    def _record_end(self):
        pass

    def _replay(self):
        self._graph.replay()

    def _eligible(self, x):
        if not x.is_cuda:
            return "batch not on a HIP device"
        if not self.model.training:
            return "model in eval mode"
        if getattr(self.model, "grad_sync", None) is not None:
            return "gradient synchronisation across ranks is issued from autograd hooks"
        if not isinstance(self.optimizer, FusedAdam):
            return f"{type(self.optimizer).__name__} has no captured form"
        return None

    def _shape_key(self, ts):
        """What a capture is valid for: the batch's shapes AND the device pointers the captured kernels were given -- the flat
        parameter / gradient buffers and every parameter view.  model.to() / .float() / flatten_parameters() or a `p.data`
        reassignment re-home them; a replay would then keep training the old memory."""
        m = self.model
        fp, fg = getattr(m, "_flat_param", None), getattr(m, "_flat_grad", None)
        ptrs = (fp.data_ptr() if fp is not None else 0, fg.data_ptr() if fg is not None else 0) + tuple(p.data_ptr() for p in self.params)
        return tuple((tuple(t.shape), t.dtype, t.device) for t in ts), ptrs

    def invalidate(self):
        """Drop the captured graph (the next eligible call captures again)."""
        self._graph = self._static = self._loss = self._grads = self._key = self._cap_token = None

    def __del__(self):
        try:
            if self._stream is not None:
                _lib.unpin_splitk_workspace(self._stream)
        except Exception:      # noqa: BLE001 -- interpreter shutdown
            pass

    def _capture(self, batch):
        dev = batch[0].device
        for p in self.params:
            p.grad = None                        # the captured backward creates the gradients (graph-owned memory)
        # no warm-up iterations on the capture stream (they would move the weights): the eager calls before this one have sized
        # everything that is allocated lazily, and a capture records launches without running them
        static = [torch.empty_like(t) for t in batch]
        for s, t in zip(static, batch):
            s.copy_(t)
        self.model.invalidate_packed()           # the capture must contain the weight packing: every replay follows an update
        if self._stream is None:
            # the engine registers a split-K scratch per stream on first use and evicts the least recently used one beyond eight
            # streams (with a device synchronisation): neither may happen inside a capture, nor to a buffer a graph points into
            self._stream = torch.cuda.Stream(device=dev)
            self._scratch = _lib.pin_splitk_workspace(self._stream)
        g = torch.cuda.CUDAGraph()
        if self._one is None or self._one.device != dev:
            self._one = torch.ones((), device=dev, dtype=torch.float32)      # dL/dL: a persistent tensor instead of autograd's own
        torch.cuda.synchronize(dev)
        with torch.cuda.graph(g, stream=self._stream, capture_error_mode="thread_local"):
            self._record_begin()
            try:
                out = self.model(static[0])
                loss = masked_l1_loss(out, static[1], static[2], static[3], static[4], grad_sync=getattr(self.model, "grad_sync", None))
                loss.backward(self._one)
                self.optimizer.capture_step()
                sloss = loss.detach()
            finally:
                self._record_end()
        self._graph, self._static, self._loss = g, static, sloss
        self._grads = [p.grad for p in self.params]
        self._key = self._shape_key(batch)
        self._cap_token = self.optimizer._cap

    def __call__(self, x, y, mask, mean, std):
        """-> the iteration's loss as a 0-dim device tensor.  After a replay it is the graph's own output buffer: read it (or
        add it into an accumulator on the same stream) before the next call."""
        self._calls += 1
        batch = (x, y, mask, mean, std)
        why = self._eligible(x)
        if why is None and not all(isinstance(t, torch.Tensor) and t.is_cuda for t in batch):
            why = "target / mask / mean / std not on the HIP device"
        if why is None and self._calls <= self.warmup:
            why = "warm-up"
        if why is None and self._graph is not None and self.optimizer._cap is not self._cap_token:
            self.invalidate()                    # optimizer.load_state_dict / a new optimizer state: capture again
        if why is None and self._graph is not None:
            key = self._shape_key(batch)
            if key[1] != self._key[1]:
                self.invalidate()                # parameters re-homed since the capture: capture again (below)
            elif key[0] != self._key[0]:
                why = "batch shape differs from the captured one"
        if why is not None:
            self.why_eager = why
            return self._eager(*batch)
        with _lib.device_of(x):
            if self._graph is None:
                if self.optimizer._cap is None:
                    # FusedAdam.capture_prepare checks the flat layout of parameters, gradients and moments: it needs the
                    # gradients of an eager iteration still in place
                    keep, self.keep_grads = self.keep_grads, True
                    loss = self._eager(*batch)   # this call's iteration, eagerly, leaving p.grad in place
                    self.keep_grads = keep
                    ok = self.optimizer.capture_prepare()
                    if not keep:
                        for p in self.params:
                            p.grad = None
                    if not ok:
                        self.warmup = 1 << 62    # never try again
                        self.why_eager = "FusedAdam is not on its one-launch flat path"
                        return loss
                    self.why_eager = "capture preparation"
                    return loss
                self._capture(batch)
            else:
                # the new batch into the static input buffers: one launch for all of them (five blit kernels were 55 us at the head
                # of every step, in front of the first convolution)
                _lib.copy_segments([(s, t) for s, t in zip(self._static, batch) if s.data_ptr() != t.data_ptr()])
            self.optimizer.advance()
            self._replay()
            self.replays += 1
            self.why_eager = None
            if self.keep_grads:
                for p, gbuf in zip(self.params, self._grads):
                    p.grad = gbuf
            else:
                for p in self.params:
                    p.grad = None
        return self._loss
