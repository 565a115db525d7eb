"""TEST-ONLY collective shim: lets two processes that SHARE one GPU run resdepth_amd's data-parallel
code path (RCCL refuses two ranks on the same device, and torch's gloo backend may or may not take
device tensors in a given build).

`install()` replaces torch.distributed's all_reduce / broadcast / reduce / all_gather for DEVICE tensors
by host-staged versions over the (CPU) gloo group, keeping the stream semantics of an RCCL collective:

  * the device->host copy is enqueued on the stream that is CURRENT at the call (exactly where RCCL would
    read the buffer), and only that stream is synchronised -- a producer running on another stream that was
    not ordered before the call delivers stale data here just as it would to RCCL;
  * the host->device copy of the result is enqueued on the same stream; `async_op=True` returns a handle
    whose `wait()` makes the then-current stream wait for that copy (RCCL's Work.wait()).

Nothing in resdepth_amd imports this module.
"""
import torch
import torch.distributed as dist

_ORIG = {}


class _Handle:
    def __init__(self, stream):
        self.ev = torch.cuda.Event()
        self.ev.record(stream)

    def wait(self, timeout=None):
        torch.cuda.current_stream().wait_event(self.ev)
        return True

    def is_completed(self):
        return self.ev.query()


def _to_host(t):
    s = torch.cuda.current_stream(t.device)
    h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    h.copy_(t, non_blocking=True)        # on the current stream
    s.synchronize()                      # ONLY the current stream
    return h


def _back(t, h, async_op):
    t.copy_(h, non_blocking=True)        # on the current stream; pinned source
    hd = _Handle(torch.cuda.current_stream(t.device))
    hd.keep = h
    return hd if async_op else None


def install():
    if _ORIG:
        return
    for name in ("all_reduce", "broadcast", "reduce", "all_gather"):
        _ORIG[name] = getattr(dist, name)

    def all_reduce(tensor, op=dist.ReduceOp.SUM, group=None, async_op=False):
        if not tensor.is_cuda:
            return _ORIG["all_reduce"](tensor, op=op, group=group, async_op=async_op)
        h = _to_host(tensor)
        _ORIG["all_reduce"](h, op=op, group=group)
        return _back(tensor, h, async_op)

    def broadcast(tensor, src=0, group=None, async_op=False, **kw):
        if not tensor.is_cuda:
            return _ORIG["broadcast"](tensor, src=src, group=group, async_op=async_op, **kw)
        h = _to_host(tensor)
        _ORIG["broadcast"](h, src=src, group=group)
        return _back(tensor, h, async_op)

    def reduce(tensor, dst=0, op=dist.ReduceOp.SUM, group=None, async_op=False, **kw):
        if not tensor.is_cuda:
            return _ORIG["reduce"](tensor, dst=dst, op=op, group=group, async_op=async_op, **kw)
        h = _to_host(tensor)
        _ORIG["reduce"](h, dst=dst, op=op, group=group)
        if dist.get_rank(group) != dst:
            return _Handle(torch.cuda.current_stream(tensor.device)) if async_op else None
        return _back(tensor, h, async_op)

    def all_gather(tensor_list, tensor, group=None, async_op=False):
        if not tensor.is_cuda:
            return _ORIG["all_gather"](tensor_list, tensor, group=group, async_op=async_op)
        h = _to_host(tensor)
        hl = [torch.empty_like(h) for _ in tensor_list]
        _ORIG["all_gather"](hl, h, group=group)
        for d, s in zip(tensor_list, hl):
            d.copy_(s)
        return _Handle(torch.cuda.current_stream(tensor.device)) if async_op else None

    dist.all_reduce, dist.broadcast, dist.reduce, dist.all_gather = all_reduce, broadcast, reduce, all_gather


def device_tensors_supported() -> bool:
    """Does this torch build's gloo backend take device tensors?  (Local check, no communication on failure: the
    backend validates its arguments before it touches the wire; on success both ranks complete one tiny all-reduce.)"""
    t = torch.ones(4, device="cuda")
    try:
        dist.all_reduce(t)
        torch.cuda.synchronize()
    except Exception:  # noqa: BLE001
        return False
    return bool(float(t[0]) == dist.get_world_size())
