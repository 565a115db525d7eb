"""Fused Adam over the model's flat parameter buffer (torch.optim.Adam as configured by
lib/utils.py:329-331: coupled L2 weight decay, betas (0.9, 0.999), eps 1e-8).

`FusedAdam` is a torch.optim.Optimizer with torch.optim.Adam's param_group keys and per-parameter
state ('step', 'exp_avg', 'exp_avg_sq'), so optimizer state_dicts are interchangeable with the
reference's checkpoints (lib/Trainer.py:145-157).  When all parameters (and their .grad) are views
into one flat buffer -- which resdepth_amd.UNet arranges -- the whole step is ONE kernel launch;
otherwise it falls back to one launch per tensor (still the HIP kernel, never torch math).
"""
from __future__ import annotations

import math

import torch

from . import _lib, ops


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0 or eps < 0 or weight_decay < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1:
            raise ValueError("invalid Adam hyper-parameter")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False,
                        foreach=None, capturable=False, differentiable=False, fused=None)
        super().__init__(params, defaults)
        self._flat_state = {}      # group index -> (flat_ptr, m, v)
        self.grad_scale = 1.0
        self._cap = None           # captured-step support (capture_prepare / capture_step / advance)

    def load_state_dict(self, state_dict):
        """torch.optim.Optimizer.load_state_dict replaces state[p]['exp_avg'|'exp_avg_sq'] by the loaded tensors; the flat
        moment buffers of a previous step would silently keep being used (and the loaded moments ignored), so drop
        them: the next step() copies the loaded moments into fresh flat buffers and re-points the state views."""
        super().load_state_dict(state_dict)
        self._flat_state = {}
        self._cap = None           # a captured step holds the OLD moment buffers: resdepth_amd.graph re-captures

    def __setstate__(self, state):
        super().__setstate__(state)
        self._flat_state = {}
        self._cap = None

    def state_dict(self):
        self._sync_steps()
        return super().state_dict()

    # ---- captured-step support: the step as a fixed launch whose scalars live in device memory --------------------------
    # resdepth_amd.graph.GraphedTrainStep captures forward + loss + backward + capture_step() into one HIP graph.  What changes
    # from step to step on the HOST side of torch.optim.Adam -- the step count in the bias corrections, a scheduler's learning
    # rate, grad_scale -- is written by advance() into an 8-float device block before every replay (rd_adam_step_dev).
    _RING = 64

    def capture_prepare(self):
        """Outside any capture.  True if every group is on the one-launch flat path with its moments allocated (after at least
        one eager step with all gradients present); allocates the scalar blocks."""
        self._sync_steps()
        groups = []
        for gi, group in enumerate(self.param_groups):
            params = list(group["params"])
            if not params:
                continue
            if group.get("amsgrad") or group.get("maximize") or any(p.grad is None or not p.is_cuda for p in params):
                return False
            flat = self._flat_state.get(gi)
            pr = self._flat_range([p.data for p in params])
            gr = self._flat_range([p.grad for p in params])
            if flat is None or pr is None or gr is None or pr[0] != flat[0] or gr[1] != flat[1].numel() or \
                    not self._same_layout(params, pr, gr):
                return False
            dev = params[0].device
            groups.append(dict(gi=gi, t=float(self.state[params[0]]["step"]), dev=torch.zeros(8, device=dev, dtype=torch.float32),
                               ring=torch.zeros(self._RING, 8, dtype=torch.float32).pin_memory(), events=[None] * self._RING, slot=0))
        if not groups:
            return False
        self._cap = groups
        return True

    def capture_step(self):
        """Inside the capture (gradients of this very capture in place): one rd_adam_step_dev per group."""
        for c in self._cap:
            group = self.param_groups[c["gi"]]
            params = list(group["params"])
            flat = self._flat_state[c["gi"]]
            pr = self._flat_range([p.data for p in params])
            gr = self._flat_range([p.grad for p in params])
            if pr is None or gr is None or pr[0] != flat[0] or gr[1] != flat[1].numel() or not self._same_layout(params, pr, gr):
                raise RuntimeError("FusedAdam.capture_step: the gradients of the captured backward are not one flat buffer in the "
                                   "parameters' layout")
            with _lib.device_of(params[0]):
                ops.adam_step_dev(_as_flat(params[pr[2]].data, gr[1]), _as_flat(params[gr[2]].grad, gr[1]), flat[1], flat[2], c["dev"])

    def advance(self):
        """Before every replay (stream-ordered in front of it): count the step, refresh the scalar blocks from the param_groups
        as they are NOW, tell the packed-weight caches that the parameters are about to change."""
        import numpy as np
        for c in self._cap:
            group = self.param_groups[c["gi"]]
            b1, b2 = group["betas"]
            c["t"] += 1.0
            t = c["t"]
            bc1 = 1.0 - b1 ** t
            bc2 = 1.0 - b2 ** t
            i = c["slot"]
            c["slot"] = (i + 1) % self._RING
            if c["events"][i] is not None:
                c["events"][i].synchronize()          # the copy that last used this pinned slot has left it
            c["ring"][i].copy_(torch.from_numpy(np.array([1.0 - b1, b2, 1.0 - b2, group["eps"], group["weight_decay"], group["lr"] / bc1,
                                                          math.sqrt(bc2), self.grad_scale], dtype=np.float64).astype(np.float32)))
            with _lib.device_of(c["dev"]):
                c["dev"].copy_(c["ring"][i], non_blocking=True)
                ev = c["events"][i] or torch.cuda.Event()
                ev.record()
                c["events"][i] = ev
            for p in group["params"]:
                _lib.bump_param_generation(p.data_ptr())

    def _sync_steps(self):
        """state[p]['step'] (torch.optim.Adam's per-parameter CPU counters) <- the captured path's python counter."""
        if self._cap:
            for c in self._cap:
                for p in self.param_groups[c["gi"]]["params"]:
                    st = self.state[p]
                    if "step" in st and float(st["step"]) != c["t"]:
                        st["step"].fill_(c["t"])

    @staticmethod
    def _flat_range(tensors):
        """(base_ptr, numel, index of the tensor at the base) if the tensors tile one contiguous fp32 range -- in ANY order:
        resdepth_amd.UNet lays its flat buffers out by gradient COMPLETION order, not by parameters() order
        (UNet._flat_layout) -- else None."""
        order = sorted(range(len(tensors)), key=lambda i: tensors[i].data_ptr())
        base = tensors[order[0]].data_ptr()
        off = 0
        for i in order:
            t = tensors[i]
            if t.dtype != torch.float32 or not t.is_contiguous() or t.data_ptr() != base + 4 * off:
                return None
            off += t.numel()
        return base, off, order[0]

    @staticmethod
    def _same_layout(params, pr, gr):
        """Do the gradients sit at the same offsets of their flat range as the parameters of theirs?"""
        return all(p.grad.data_ptr() - gr[0] == p.data_ptr() - pr[0] for p in params)

    def _ensure_state(self, group, gi):
        params = [p for p in group["params"]]
        fr = self._flat_range([p.data for p in params])
        dev = params[0].device
        have = self._flat_state.get(gi)
        if fr is not None:
            total = fr[1]
            if have is None or have[0] != fr[0] or have[1].numel() != total:
                m = torch.zeros(total, device=dev, dtype=torch.float32)
                v = torch.zeros(total, device=dev, dtype=torch.float32)
                for p in params:
                    st = self.state[p]
                    n, off = p.numel(), (p.data_ptr() - fr[0]) // 4      # the moments mirror the parameters' layout
                    if "exp_avg" in st:          # e.g. after load_state_dict
                        m[off:off + n].copy_(st["exp_avg"].reshape(-1))
                        v[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
                    st["exp_avg"] = m[off:off + n].view(p.shape)
                    st["exp_avg_sq"] = v[off:off + n].view(p.shape)
                    st.setdefault("step", torch.tensor(0.0))
                self._flat_state[gi] = (fr[0], m, v)
            return self._flat_state[gi]
        for p in params:
            st = self.state[p]
            if "exp_avg" not in st:
                st["step"] = torch.tensor(0.0)
                st["exp_avg"] = torch.zeros_like(p.data)
                st["exp_avg_sq"] = torch.zeros_like(p.data)
        return None

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._sync_steps()
        for gi, group in enumerate(self.param_groups):
            params = [p for p in group["params"]]
            if not params:
                continue
            with _lib.device_of(params[0]):
                self._step_group(gi, group, params)
        if self._cap:                      # an eager step between replays (a ragged batch): the captured counter follows
            for c in self._cap:
                ps = self.param_groups[c["gi"]]["params"]
                if ps and "step" in self.state[ps[0]]:
                    c["t"] = float(self.state[ps[0]]["step"])
        return loss

    def _step_group(self, gi, group, params):
        if any(not p.is_cuda for p in params):
            raise RuntimeError("resdepth_amd.FusedAdam: parameters must live on a HIP device (no CPU fallback)")
        if group.get("amsgrad") or group.get("maximize"):
            raise NotImplementedError("FusedAdam: amsgrad / maximize are not implemented")
        if any(p.grad is None for p in params):
            # torch.optim.Adam skips parameters without gradient; the flat kernel cannot
            active = [p for p in params if p.grad is not None]
            flat = None
        else:
            active = params
            flat = self._ensure_state(group, gi)
        b1, b2 = group["betas"]
        lr, eps, wd = group["lr"], group["eps"], group["weight_decay"]
        if flat is not None:
            gr = self._flat_range([p.grad for p in params])
        else:
            gr = None
            self._ensure_state_per_tensor(active)
        # step counter (python float tensors like torch.optim.Adam's default path)
        for p in active:
            self.state[p]["step"] += 1
        if not active:
            return
        t = float(self.state[active[0]]["step"])
        bc1 = 1.0 - b1 ** t
        bc2 = 1.0 - b2 ** t
        step_size = lr / bc1
        bc2_sqrt = math.sqrt(bc2)
        pr = self._flat_range([p.data for p in params]) if flat is not None and gr is not None else None
        if pr is not None and gr[1] == flat[1].numel() and pr[0] == flat[0] and self._same_layout(params, pr, gr):
            total = gr[1]
            pflat = _as_flat(params[pr[2]].data, total)
            gflat = _as_flat(params[gr[2]].grad, total)
            ops.adam_step(pflat, gflat, flat[1], flat[2], b1, b2, eps, wd, step_size, bc2_sqrt, self.grad_scale)
            # every parameter pointer of the group, not only the flat buffer's first: a model's pack key reads the generations
            # of ITS OWN parameters, and one group may span several models' (contiguously allocated) flat buffers
            for p in params:
                _lib.bump_param_generation(p.data_ptr())
        else:
            for p in active:
                st = self.state[p]
                g = p.grad.contiguous()
                ops.adam_step(p.data.view(-1) if p.data.is_contiguous() else p.data, g.view(-1),
                              st["exp_avg"].view(-1), st["exp_avg_sq"].view(-1), b1, b2, eps, wd, step_size,
                              bc2_sqrt, self.grad_scale)
            # `.data` writes do not bump Parameter._version: tell the packed-weight caches of the models that own these
            # tensors (UNet._current_pack_key reads the generation of every parameter pointer) -- and only those
            for p in active:
                _lib.bump_param_generation(p.data_ptr())

    def _ensure_state_per_tensor(self, params):
        for p in params:
            st = self.state[p]
            if "exp_avg" not in st:
                st["step"] = torch.tensor(0.0)
                st["exp_avg"] = torch.zeros_like(p.data)
                st["exp_avg_sq"] = torch.zeros_like(p.data)


class FusedSGD(torch.optim.Optimizer):
    """torch.optim.SGD as the reference configures it (lib/utils.py:332-334: lr + coupled weight decay, momentum 0), with
    torch's momentum / dampening / nesterov options and state key ('momentum_buffer') so optimizer state_dicts are
    interchangeable.  One launch over the flat parameter buffer when parameters and gradients are views of flat
    buffers (resdepth_amd.UNet arranges that), else one launch per tensor -- always the HIP kernel."""

    def __init__(self, params, lr=1e-3, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False):
        if lr < 0 or momentum < 0 or weight_decay < 0:
            raise ValueError("invalid SGD hyper-parameter")
        if nesterov and (momentum <= 0 or dampening != 0):
            raise ValueError("Nesterov momentum requires a momentum and zero dampening")      # torch's own message
        defaults = dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay, nesterov=nesterov,
                        maximize=False, foreach=None, differentiable=False, fused=None)
        super().__init__(params, defaults)
        self._flat_state = {}      # group index -> (flat_ptr, momentum buffer or None, steps taken)
        self.grad_scale = 1.0

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._flat_state = {}

    def __setstate__(self, state):
        super().__setstate__(state)
        self._flat_state = {}

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            params = [p for p in group["params"]]
            if not params:
                continue
            with _lib.device_of(params[0]):
                self._step_group(gi, group, params)
        return loss

    def _step_group(self, gi, group, params):
        if any(not p.is_cuda for p in params):
            raise RuntimeError("resdepth_amd.FusedSGD: parameters must live on a HIP device (no CPU fallback)")
        if group.get("maximize"):
            raise NotImplementedError("FusedSGD: maximize is not implemented")
        lr, wd, mom = group["lr"], group["weight_decay"], group["momentum"]
        damp, nest = group["dampening"], group["nesterov"]
        pr = FusedAdam._flat_range([p.data for p in params])
        gr = FusedAdam._flat_range([p.grad for p in params]) if all(p.grad is not None for p in params) else None
        # mixed momentum state (some parameters have a buffer, others do not: partial load_state_dict, steps taken while some
        # gradients were None): torch starts the missing buffers as clone(grad), which one `first` flag for the whole group
        # cannot express -- this step goes per tensor (below), after which every parameter has a buffer
        n_buf = sum(1 for p in params if self.state[p].get("momentum_buffer") is not None) if mom != 0 else 0
        mixed = 0 < n_buf < len(params)
        if pr is not None and gr is not None and gr[1] == pr[1] and not mixed and FusedAdam._same_layout(params, pr, gr):
            total = pr[1]
            have = self._flat_state.get(gi)
            buf, first = None, False
            if mom != 0:
                if have is None or have[0] != pr[0] or have[1] is None or have[1].numel() != total:
                    buf = torch.zeros(total, device=params[0].device, dtype=torch.float32)
                    # first = "no parameter has a momentum buffer yet" (torch initialises it with the gradient);
                    # a loaded / per-tensor state is copied into the flat buffer
                    first = not any("momentum_buffer" in self.state[p] and self.state[p]["momentum_buffer"] is not None
                                    for p in params)
                    for p in params:
                        st, n, off = self.state[p], p.numel(), (p.data_ptr() - pr[0]) // 4
                        if st.get("momentum_buffer") is not None:
                            buf[off:off + n].copy_(st["momentum_buffer"].reshape(-1))
                        st["momentum_buffer"] = buf[off:off + n].view(p.shape)
                    self._flat_state[gi] = (pr[0], buf)
                else:
                    buf = have[1]
            ops.sgd_step(_as_flat(params[pr[2]].data, total), _as_flat(params[gr[2]].grad, total), buf, lr, wd, mom, damp, nest,
                         first, self.grad_scale)
            for p in params:                  # every pointer of the group (see FusedAdam._step_group)
                _lib.bump_param_generation(p.data_ptr())
            return
        for p in params:
            if p.grad is None:
                continue                      # torch.optim.SGD skips parameters without gradient
            st = self.state[p]
            buf, first = None, False
            if mom != 0:
                if st.get("momentum_buffer") is None:
                    st["momentum_buffer"] = torch.zeros_like(p.data, memory_format=torch.contiguous_format)
                    first = True
                buf = st["momentum_buffer"].view(-1)
            g = p.grad.contiguous()
            ops.sgd_step(p.data.view(-1) if p.data.is_contiguous() else p.data, g.view(-1), buf, lr, wd, mom, damp, nest,
                         first, self.grad_scale)
            _lib.bump_param_generation(p.data_ptr())


def get_optimizer(cfg, model, logger=None):
    """lib/utils.py:318-340 with the fused optimizers: cfg.optimizer.{name, learning_rate, weight_decay}; 'Adam' ->
    FusedAdam, 'SGD' -> FusedSGD (torch defaults for everything else).  Unknown names log the reference's message and,
    like the reference (which then hits an unbound local), raise."""
    name = cfg.optimizer.name
    if name == "Adam":
        return FusedAdam(model.parameters(), lr=cfg.optimizer.learning_rate, weight_decay=cfg.optimizer.weight_decay)
    if name == "SGD":
        return FusedSGD(model.parameters(), lr=cfg.optimizer.learning_rate, weight_decay=cfg.optimizer.weight_decay)
    msg = f"{name} optimizer is not implemented. Choose among ['Adam', 'SGD'].\n"
    if logger:
        logger.error(msg)
    else:
        print(f"ERROR: {msg}")
    raise UnboundLocalError("local variable 'optimizer' referenced before assignment")


def _as_flat(first: torch.Tensor, total: int) -> torch.Tensor:
    """A 1-D view of `total` fp32 elements starting at `first`'s storage position."""
    return torch.as_strided(first, (total,), (1,), first.storage_offset()) if first.storage_offset() + total <= \
        first.untyped_storage().nbytes() // 4 else _raise("flat range exceeds storage")


def _raise(msg):
    raise RuntimeError(msg)
