"""ctypes binding of libresdepth_hip.so (the C ABI declared in include/resdepth_hip.h).

The product path has NO CPU fallback: if the shared library is missing, or a tensor is not
on a HIP device, the call raises.  Build with `resdepth_amd/csrc/build.sh` (or
`python -c "import __graft_entry__ as g; g.build()"`).
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# RESDEPTH_HIP_LIB: alternative build of the same library (kernel diagnosis builds)
# RD_MFMA: arithmetic of the matrix-pipe kernels, read by the LIBRARY at load time (csrc/rd_runtime.hip) -- one shared object:
#   split2h  two fp16 terms / three products where the operands carry magnitude slots (include/resdepth_hip.h: rd_quant_next)
#   split3   three bf16 terms / six products everywhere
#   f32      v_mfma_f32_32x32x2_f32 (exact)
# and switchable in-process: tune_set("mfma_products", 3 | 6), tune_set("mfma_f32", 0 | 1).
MFMA_MODE = os.environ.get("RD_MFMA", "")
if MFMA_MODE not in ("", "f32", "split3", "split2h"):
    raise RuntimeError(f"RD_MFMA={MFMA_MODE!r}: expected f32, split3 or split2h")
LIB_PATH = os.environ.get("RESDEPTH_HIP_LIB") or os.path.join(_HERE, "libresdepth_hip.so")
AMAX_WORDS = 512     # RD_AMAX_SLOT_BYTES / 4: int32 elements of one magnitude slot (sixteen words on sixteen 128-byte lines)

_lib = None
_lock = threading.Lock()

P = C.c_void_p
I = C.c_int
F = C.c_float
D = C.c_double
LL = C.c_longlong
SZ = C.c_size_t

# name -> (restype, argtypes); mirrors include/resdepth_hip.h one to one
SIGNATURES = {
    "rd_version": (I, []),
    "rd_mfma_products": (I, []),
    "rd_plan_begin": (I, [P, P]),
    "rd_plan_event_record": (I, [P]),
    "rd_plan_event_wait": (I, [P, I]),
    "rd_plan_segment": (I, []),
    "rd_plan_poison": (I, [C.c_char_p]),
    "rd_plan_end": (P, [P, P]),
    "rd_plan_replay": (I, [P, I, P, P]),
    "rd_plan_free": (I, [P]),
    "rd_plan_dump": (I, [P]),
    "rd_quant_next": (I, [P, P, P, P]),
    "rd_quant_next_img": (I, [P, P, P, P, I]),
    "rd_amax": (I, [P, LL, P, P]),
    "rd_zero": (I, [P, SZ, P]),
    "rd_copy_segments": (I, [P, P, P, I, P]),
    "rd_last_error_string": (C.c_char_p, []),
    "rd_pack_conv3x3_weight": (I, [P, P, P, I, I, P]),
    "rd_pack_item_pieces": (LL, [I, I, I, I]),
    "rd_set_splitk_workspace": (I, [P, SZ, P]),
    "rd_pack_item_tiles": (LL, [I, I, I]),
    "rd_pack_weights_fused": (I, [P, I, LL, LL, P]),
    "rd_pack_conv3x3_weight_folded": (I, [P, P, P, I, I, P]),
    "rd_pack_convt2x2_weight": (I, [P, P, P, I, I, P]),
    "rd_conv3x3_fwd": (I, [P, P, P, I, I, I, I, I, P]),
    "rd_conv3x3_fwd_stats_ws_bytes": (SZ, [I, I, I, I, I]),
    "rd_conv3x3_fwd_stats": (I, [P, P, P, P, I, I, I, I, I, P, SZ, P]),
    "rd_conv3x3_fwd_bn": (I, [P, P, P, D, F, F, P, P, P, P, P, I, I, I, I, I, P, SZ, P]),
    "rd_conv3x3_fwd_act": (I, [P, P, P, F, P, P, I, I, I, I, I, P]),
    "rd_conv3x3_bwd_data": (I, [P, P, P, I, I, I, I, I, P]),
    "rd_conv3x3_bwd_weight_ws_bytes": (SZ, [I, I, I, I, I]),
    "rd_conv3x3_bwd_weight": (I, [P, P, P, I, I, I, I, I, P, SZ, P]),
    "rd_conv3x3_first_fwd": (I, [P, P, P, I, I, I, I, I, P]),
    "rd_conv3x3_first_fwd_stats_ws_bytes": (SZ, [I, I, I, I, I]),
    "rd_conv3x3_first_fwd_stats": (I, [P, P, P, P, I, I, I, I, I, P, SZ, P]),
    "rd_conv3x3_first_fwd_bn": (I, [P, P, P, D, F, F, P, P, P, P, P, I, I, I, I, I, P, SZ, P]),
    "rd_conv3x3_first_fwd_act_available": (I, [I, I, I, I, I]),
    "rd_conv3x3_first_fwd_act": (I, [P, P, P, P, P, P, F, P, P, P, I, I, I, I, I, P]),
    "rd_conv3x3_first_bwd_weight_ws_bytes": (SZ, [I, I, I, I, I]),
    "rd_tail_available": (I, [I, I]),
    "rd_tail_compose": (I, [P, P, P, P, P, P, P, I, I, P]),
    "rd_conv3x3_last_fwd_tail": (I, [P, P, P, P, P, F, P, P, P, P, P, P, I, P, I, I, I, I, P]),
    "rd_conv3x3_last_bwd_tail_blocks": (I, [I, I, I]),
    "rd_conv3x3_last_bwd_tail_fused": (I, [P, P, P, P, P, F, P, P, P, P, P, SZ, P, I, I, I, I, P]),
    "rd_tail_wl_finish": (I, [P, I, P, P, P, P, P, I, I, P]),
    "rd_conv3x3_last_bwd_weight_tail_ws_bytes": (SZ, [I, I, I, I]),
    "rd_conv3x3_last_bwd_weight_tail": (I, [P, P, P, P, P, F, P, P, P, P, P, P, P, I, I, I, I, I, P, SZ, P]),
    "rd_convt_last_bwd_weight_ws_bytes": (SZ, [I, I, I, I]),
    "rd_convt_last_bwd_weight": (I, [P, P, P, P, P, I, I, I, I, I, P, SZ, P]),
    "rd_convt_last_bwd_weight_bn": (I, [P, P, P, P, P, F, P, P, P, P, P, I, I, I, I, I, P, SZ, P]),
    "rd_tail_t16": (I, [P, P, P, P, P, F, P, P, P, LL, I, P]),
    "rd_convt_last_bwd_data": (I, [P, P, P, I, I, I, I, P, P, P, P, P, F, P, P, SZ, P, P]),
    "rd_conv3x3_first_bwd_weight_bn_available": (I, [I, I, I, I, I]),
    "rd_conv3x3_first_bwd_weight_bn": (I, [P, P, P, P, P, P, F, P, P, P, P, P, D, I, P, P, P, I, I, I, I, I, P, SZ, P]),
    "rd_conv3x3_first_bwd_weight": (I, [P, P, P, I, I, I, I, I, P, SZ, P]),
    "rd_conv3x3_last_fwd": (I, [P, P, P, P, I, P, I, I, I, I, P]),
    "rd_conv3x3_last_bwd_data": (I, [P, P, P, I, I, I, I, P]),
    "rd_conv3x3_last_bwd_weight_ws_bytes": (SZ, [I, I, I, I]),
    "rd_conv3x3_last_bwd_weight": (I, [P, P, P, P, I, I, I, I, P, SZ, P]),
    "rd_convt2x2_fwd": (I, [P, P, P, P, P, I, I, I, I, I, P]),
    "rd_convt2x2_fwd_bnskip": (I, [P, P, P, P, P, P, P, P, F, P, P, I, I, I, I, I, P]),
    "rd_convt2x2_bwd_data": (I, [P, P, P, I, I, I, I, I, P]),
    "rd_convt2x2_bwd_weight_ws_bytes": (SZ, [I, I, I, I, I]),
    "rd_convt2x2_bwd_weight": (I, [P, P, P, I, I, I, I, I, P, SZ, P]),
    "rd_pack_conv1x1_weight": (I, [P, P, P, I, I, P]),
    "rd_packed_weight_bytes": (SZ, [I, I, I]),
    "rd_conv1x1_fwd": (I, [P, P, P, LL, I, I, P]),
    "rd_conv1x1_bwd_data": (I, [P, P, P, LL, I, I, P]),
    "rd_conv1x1_bwd_weight_ws_bytes": (SZ, [LL, I, I]),
    "rd_conv1x1_bwd_weight": (I, [P, P, P, LL, I, I, P, SZ, P]),
    "rd_upsample2x_add_fwd": (I, [P, P, P, P, I, I, I, I, P]),
    "rd_upsample2x_bwd": (I, [P, P, I, I, I, I, P]),
    "rd_channel_sum_ws_bytes": (SZ, [LL, I]),
    "rd_channel_sum": (I, [P, P, LL, I, P, SZ, P]),
    "rd_bn_stats_ws_bytes": (SZ, [LL, I]),
    "rd_bn_stats_partial": (I, [P, P, LL, I, P, SZ, P]),
    "rd_bn_stats_finalize": (I, [P, D, F, F, P, P, P, P, P, I, P]),
    "rd_bn_eval_stats": (I, [P, P, F, P, P, I, P]),
    "rd_bn_act_pool_fwd": (I, [P, P, P, P, P, F, P, P, P, P, P, I, I, I, I, P]),
    "rd_bn_act_bwd_ws_bytes": (SZ, [I, I, I, I]),
    "rd_bn_bwd_part_floats": (SZ, [LL, I]),
    "rd_conv3x3_bwd_data_bnstats": (I, [P, P, P, I, I, I, I, I, P, P, P, P, P, F, P, I, P, SZ, P, P]),
    "rd_convt2x2_bwd_data_bnstats": (I, [P, P, P, I, I, I, I, I, P, P, P, P, P, F, P, P, SZ, P, P]),
    "rd_conv3x3_last_bwd_data_bnstats": (I, [P, P, P, I, I, I, I, P, P, P, P, P, F, P, P, SZ, P, P]),
    "rd_bn_bwd_stats_finalize": (I, [P, I, P, I, I, P, P, P, P, P]),
    "rd_bn_act_bwd_reduce": (I, [P, P, P, P, P, F, P, P, P, P, P, P, P, P, I, I, I, I, P, SZ, P]),
    "rd_bn_act_bwd_apply": (I, [P, P, P, P, P, F, P, P, P, P, P, D, I, P, P, P, I, I, I, I, P]),
    "rd_masked_l1_ws_bytes": (SZ, [LL]),
    "rd_masked_l1_partial": (I, [P, P, P, P, P, P, I, LL, P, SZ, P]),
    "rd_masked_l1_finish": (I, [P, P, P, P, P, P, D, P, P, P, I, LL, P]),
    "rd_adam_step": (I, [P, P, P, P, LL, D, D, F, F, F, F, F, P]),
    "rd_adam_step_dev": (I, [P, P, P, P, LL, P, P]),
    "rd_sgd_step": (I, [P, P, P, LL, F, F, F, F, I, I, F, P]),
    "rd_blend_accumulate": (I, [P, P, P, P, P, I, I, I, P, I, I, P]),
    "rd_host_register": (I, [P, SZ]),
    "rd_host_unregister": (I, [P]),
    "rd_copy_to_host_async": (I, [P, P, SZ, P]),
    "rd_patch_sums": (I, [P, LL, P, I, P, I, I, I, F, I, P, P]),
    "rd_assemble_patches": (I, [P, P, P, LL, P, I, P, P, P, F, P, F, F, I, I, I, P, P, P, P]),
    "rd_residual_stats_ws_bytes": (SZ, [LL]),
    "rd_residual_stats": (I, [P, P, P, LL, D, D, P, P, SZ, P]),
    "rd_nchw_to_nhwc": (I, [P, P, I, I, I, I, P]),
    "rd_nhwc_to_nchw": (I, [P, P, I, I, I, I, P]),
    "rd_tune_set": (I, [C.c_char_p, I]),
    "rd_tune_get": (I, [C.c_char_p, P]),
    "rd_prof_enable": (I, [I]),
    "rd_prof_reset": (I, []),
    "rd_prof_collect": (I, [P, I]),
}


class ProfEntry(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("launches", C.c_longlong), ("ms", C.c_double),
                ("flops", C.c_double), ("bytes", C.c_double)]


def load():
    """Load (once) and return the ctypes handle; raises RuntimeError if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"resdepth_amd: HIP library not built ({LIB_PATH} missing). Run resdepth_amd/csrc/build.sh; "
                "there is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)       # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        if lib.rd_version() < 106:
            raise RuntimeError(f"resdepth_amd: {LIB_PATH} is version {lib.rd_version()} (< 106): stale build -- run "
                               "resdepth_amd/csrc/build.sh")
        _lib = lib
    return _lib


def mfma_mode() -> str:
    """Arithmetic of the matrix-pipe convolution kernels in this process: "split2h" (two fp16 terms, three products per multiply
    on operands with magnitude slots), "split3" (three bf16 terms, six products) or "f32" (exact)."""
    if tune_get("mfma_f32"):
        return "f32"
    return "split2h" if load().rd_mfma_products() == 3 else "split3"


_products = None


def products() -> int:
    """3 when GEMM operands should carry magnitude slots (split2h mode, split kernels active), else 6.  Cached; tune_set drops
    the cache."""
    global _products
    if _products is None:
        _products = 3 if (load().rd_mfma_products() == 3 and not tune_get("mfma_f32")) else 6
    return _products


# ---- magnitude slots (include/resdepth_hip.h: rd_quant_next) ----------------------------------------------------------------
# A tensor that is an operand of a three-product GEMM carries its slot as the attribute `_rd_amax` (a 2 KB int32 view of a
# pool); producers draw the slot from the pool that is ACTIVE on the calling thread (`with AmaxPool(...)`: the engine's forward
# and backward), consumers read the attribute -- a tensor without one (any torch op's result, a view, .contiguous()) simply
# runs the six-product body.
_tls = threading.local()


class AmaxPool:
    """A zeroed block of magnitude slots for one pass (forward or backward) of one model: one torch.zeros per pass.
    per_image = n > 0 (inference): every take() is an ARRAY of n slots, one per image of the batch (rd_quant_next_img) -- the
    view carries `_rd_img = n`."""

    def __init__(self, device, slots: int = 64, per_image: int = 0):
        self.img = int(per_image)
        self.words = AMAX_WORDS * max(1, self.img)
        self.buf = zeros_i32(slots * self.words, device)
        self.n, self.cap = 0, slots

    def take(self):
        if self.n >= self.cap:
            return None                      # more producers than planned: the tensor goes without (six-product consumer)
        v = self.buf[self.n * self.words:(self.n + 1) * self.words]
        if self.img:
            v._rd_img = self.img
        self.n += 1
        return v

    def __enter__(self):
        self._prev = getattr(_tls, "pool", None)
        _tls.pool = self
        return self

    def __exit__(self, *exc):
        _tls.pool = self._prev
        return False


def zeros_i32(n: int, device) -> torch.Tensor:
    """torch.zeros(n, int32) with the zeroing done by a kernel of the library (rd_zero: a launch plan records it)."""
    t = torch.empty(n, dtype=torch.int32, device=device)
    zero_(t)
    return t


def copy_segments(pairs) -> None:
    """dst.copy_(src) for up to eight (dst, src) pairs of same-shaped contiguous device tensors in ONE launch on the current
    stream (rd_copy_segments); pairs the kernel cannot take (alignment, odd sizes, dtype / layout differences) go through torch."""
    fast = []
    for d, s in pairs:
        nb = d.numel() * d.element_size()
        if (d.is_cuda and s.is_cuda and d.dtype == s.dtype and d.shape == s.shape and d.is_contiguous() and s.is_contiguous()
                and nb % 16 == 0 and d.data_ptr() % 16 == 0 and s.data_ptr() % 16 == 0 and len(fast) < 8):
            fast.append((d, s, nb))
        else:
            d.copy_(s, non_blocking=True)
    if not fast:
        return
    n = len(fast)
    dst = (C.c_void_p * n)(*[d.data_ptr() for d, _, _ in fast])
    src = (C.c_void_p * n)(*[s.data_ptr() for _, s, _ in fast])
    nbs = (C.c_size_t * n)(*[nb for _, _, nb in fast])
    with device_of(fast[0][0]):
        check(load().rd_copy_segments(dst, src, nbs, n, stream_ptr()), "copy_segments")


def zero_(t: torch.Tensor) -> torch.Tensor:
    with device_of(t):
        check(load().rd_zero(t.data_ptr(), t.numel() * t.element_size(), stream_ptr()), "zero")
    return t


# ---- stream ordering through the library, so that a launch plan can record it (include/resdepth_hip.h: rd_plan_*) -----------
_plan_rec = None          # the PlanRecorder of the recording in progress (resdepth_amd/plan.py), else None


class Ev:
    """A recorded position of a stream: torch.cuda.Event + its index in the plan being recorded (-1: none)."""
    __slots__ = ("e", "idx")

    def __init__(self, stream):
        self.e = torch.cuda.Event()
        self.e.record(stream)
        self.idx = load().rd_plan_event_record(stream.cuda_stream) if _plan_rec is not None else -1


def ev_wait(stream, ev: "Ev") -> None:
    stream.wait_event(ev.e)
    if _plan_rec is not None:
        load().rd_plan_event_wait(stream.cuda_stream, ev.idx)


def wait_stream(dst, src) -> None:
    """dst waits for everything enqueued on src so far (torch's Stream.wait_stream, visible to a plan recording)."""
    ev_wait(dst, Ev(src))


def host_action(fn) -> None:
    """Something the HOST does between kernels of an iteration -- a collective of torch.distributed.  Eager: runs now.  While a
    plan is recorded: the current segment ends here and `fn` becomes what the replay loop calls after it; it is NOT run now (the
    recording executes nothing: it happens under a graph capture, whose results are discarded)."""
    if _plan_rec is None:
        fn()
    else:
        _plan_rec.add_action(fn)


def amax_slot(img_ok: bool = False):
    """A fresh slot from the active pool, or None (no pool / six-product mode).  A per-image pool serves only the producers that
    commit per image (img_ok: the three entry points of the folded inference path); everyone else's output goes without."""
    pool = getattr(_tls, "pool", None)
    if pool is None or (pool.img and not img_ok):
        return None
    return pool.take()


def slot_of(t):
    return getattr(t, "_rd_amax", None) if t is not None else None


def tag(t, slot):
    if t is not None and slot is not None:
        t._rd_amax = slot
    return t


def quant_next(a=None, b=None, out=None, out2=None):
    """rd_quant_next with tensors (None -> NULL): only worth a call when something is set.  Slot ARRAYS (views of a per-image
    pool, `_rd_img`) go through rd_quant_next_img -- the caller has made sure a, out, out2 are all of that kind."""
    if a is None and b is None and out is None and out2 is None:
        return
    args = (a.data_ptr() if a is not None else None, b.data_ptr() if b is not None else None,
            out.data_ptr() if out is not None else None, out2.data_ptr() if out2 is not None else None)
    if any(getattr(t, "_rd_img", 0) for t in (a, out, out2) if t is not None):
        check(load().rd_quant_next_img(*args, AMAX_WORDS), "quant_next_img")
    else:
        load().rd_quant_next(*args)


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().rd_last_error_string()
        raise RuntimeError(f"libresdepth_hip {what} failed (code {rc}): {msg.decode() if msg else ''}")


def stream_ptr() -> int:
    """HIP stream the next kernel is enqueued on: torch's current stream of the CURRENT device.  The engine entry points
    (UNet.forward / backward, the loss, FusedAdam.step, predict_linear_blend) run under `device_of(tensor)` so that the
    current device is the one the tensors live on."""
    return torch.cuda.current_stream().cuda_stream


def device_of(t):
    """Context manager making `t`'s device current (torch ops guard the device themselves; raw launches do not)."""
    if not t.is_cuda:
        return contextlib.nullcontext()        # the call underneath raises "no CPU fallback"
    return torch.cuda.device(t.device)


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Refuses host tensors: no CPU path exists."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("resdepth_amd: tensor is not on a HIP device; the HIP path has no CPU fallback")
    if not t.is_contiguous():
        raise RuntimeError("resdepth_amd: tensor must be contiguous")
    return t.data_ptr()


# ---- workspace: one growing scratch buffer per (device, HIP stream).  Every use is stream-ordered on the stream the
# buffer is keyed by, so two models, or the forward and the autograd thread, can only share a buffer when they are
# serialised by that stream anyway; concurrent streams (the weight-gradient side stream) get their own buffer.
_ws = {}
_ws_lock = threading.Lock()


def workspace(nbytes: int, device, slot: int = 0) -> torch.Tensor:
    dev = torch.device(device)
    index = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (index, torch.cuda.current_stream(index).cuda_stream, slot)
    with _ws_lock:
        buf = _ws.get(key)
        if buf is None or buf.numel() < nbytes:
            # the replaced buffer goes back to torch's caching allocator, which re-uses a block only in the order of the
            # stream it was allocated under -- the same stream that is current here
            buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=torch.device("cuda", index))
            _ws[key] = buf
    return buf


_splitk = {}
_splitk_clock = 0
# the most a launch uses: tiles * ranges <= 1024 partial tiles of 128 x 64 floats (csrc/rd_igemm.hip launch_nt) + 64 KB of tickets
SPLITK_BYTES = (32 << 20) + (64 << 10)
SPLITK_MAX_STREAMS = 8          # per device; a ninth stream takes over the least recently used registration


def ensure_splitk_workspace(device, nbytes: int = SPLITK_BYTES) -> None:
    """Register (once per device and stream) the split-K scratch of the 8 x 8 convolution kernel for the CURRENT stream of
    `device` (include/resdepth_hip.h: rd_set_splitk_workspace; the library keys it by (device, stream) too).  The engine
    entry points call this; direct users of the op wrappers may too -- without it those layers run the same kernel unsplit
    (same bits, fewer blocks at small batches).  At most SPLITK_MAX_STREAMS streams per device hold a buffer, so a server
    running requests on many short-lived streams does not grow device memory by one scratch per stream handle; beyond that
    the least recently used registration is evicted and its buffer re-used."""
    dev = torch.device(device)
    index = dev.index if dev.index is not None else torch.cuda.current_device()
    stream = torch.cuda.current_stream(index).cuda_stream
    key = (index, stream)
    global _splitk_clock
    with _ws_lock:
        _splitk_clock += 1
        if key in _splitk:
            if _splitk[key][1] != _PINNED:
                _splitk[key][1] = _splitk_clock
            return
        mine = [k for k in _splitk if k[0] == index and _splitk[k][1] != _PINNED]
        buf = None
        if len(mine) >= SPLITK_MAX_STREAMS:
            # evict the least recently used registration of this device and hand its scratch to the new stream: a process that
            # touched eight short-lived streams must not lose the split-K path on every later one.  The old stream's pending
            # launches still reference the buffer, so the new stream first waits for everything enqueued on the device
            # (rare: only when a ninth stream appears)
            victim = min(mine, key=lambda k: _splitk[k][1])
            with torch.cuda.device(index):
                check(load().rd_set_splitk_workspace(None, 0, victim[1]), "set_splitk_workspace")
                torch.cuda.synchronize(index)
            buf = _splitk.pop(victim)[0]
        if buf is None or buf.numel() < int(nbytes):
            buf = torch.empty(int(nbytes), dtype=torch.uint8, device=torch.device("cuda", index))
        with torch.cuda.device(index):
            check(load().rd_set_splitk_workspace(buf.data_ptr(), buf.numel(), stream), "set_splitk_workspace")
        _splitk[key] = [buf, _splitk_clock]     # kept alive while registered (the library holds the raw pointer)


_PINNED = float("inf")


def pin_splitk_workspace(stream: "torch.cuda.Stream", nbytes: int = SPLITK_BYTES):
    """A split-K scratch of its own for `stream`, outside the least-recently-used turnover of ensure_splitk_workspace: what a
    captured HIP graph needs (resdepth_amd/graph.py) -- its kernels keep the raw pointer for as long as the graph is replayed, and
    registering inside a capture could need the eviction's device synchronisation, which a capture forbids.  Call outside any
    capture; returns the buffer (the caller keeps it alive); unpin_splitk_workspace(stream) gives the registration up."""
    index = stream.device.index
    key = (index, stream.cuda_stream)
    with _ws_lock:
        old = _splitk.pop(key, None)
        # allocated UNDER `stream`: the registration zeroes the ticket area with a memset on that stream, and torch's allocator
        # only hands a stream blocks whose earlier uses are ordered before it on the SAME stream.  (r06: allocated under the
        # caller's current stream, the block could be one the previous iteration's kernels -- still running there -- had just
        # given back; they then overwrote the zeroed tickets, and the first split-K launch of the capture's replay went wrong:
        # found by the launch plan's bit-for-bit verification at batch 32, where the GPU lags the host by several iterations.)
        with torch.cuda.device(index), torch.cuda.stream(stream):
            buf = old[0] if old is not None and old[0].numel() >= int(nbytes) else \
                torch.empty(int(nbytes), dtype=torch.uint8, device=torch.device("cuda", index))
            check(load().rd_set_splitk_workspace(buf.data_ptr(), buf.numel(), stream.cuda_stream), "set_splitk_workspace")
        _splitk[key] = [buf, _PINNED]
    return buf


def unpin_splitk_workspace(stream: "torch.cuda.Stream") -> None:
    key = (stream.device.index, stream.cuda_stream)
    with _ws_lock:
        if key in _splitk and _splitk[key][1] == _PINNED:
            with torch.cuda.device(key[0]):
                check(load().rd_set_splitk_workspace(None, 0, key[1]), "set_splitk_workspace")
            _splitk.pop(key)


# ---- parameter generation counter (bumped by in-place updates done through raw pointers) -------
_param_gen = {}
_global_gen = 0
_PARAM_GEN_MAX = 1 << 16        # ~1500 models' worth of parameter pointers


def bump_param_generation(flat_ptr=None):
    """Tell every packed-weight cache that parameters were rewritten through raw pointers / `.data` (which does not bump
    `Parameter._version`).  flat_ptr = the flat buffer that changed; None = "some parameter somewhere" (per-tensor
    optimizer fallback, broadcast): invalidates every model's cache."""
    global _global_gen
    if flat_ptr is None:
        _global_gen += 1
    else:
        if len(_param_gen) >= _PARAM_GEN_MAX and flat_ptr not in _param_gen:
            # entries of freed models are never removed one by one (a raw pointer has no owner to ask): when the table is
            # full it is dropped whole and the global counter moves, which makes every live cache re-validate once
            _param_gen.clear()
            _global_gen += 1
        _param_gen[flat_ptr] = _param_gen.get(flat_ptr, 0) + 1


def param_generation(flat_ptr: int):
    return (_param_gen.get(flat_ptr, 0), _global_gen)


def generations(ptrs):
    """Raw-pointer generations of several tensors (a model's parameters): what a per-tensor optimizer step or any other
    `.data` writer that calls bump_param_generation(p.data_ptr()) leaves behind.  Does not include the global counter."""
    g = _param_gen
    return tuple(g.get(q, 0) for q in ptrs)


def global_generation() -> int:
    return _global_gen


# ---- diagnosis knobs ----------------------------------------------------------------------------
def tune_set(name: str, value: int):
    global _products
    check(load().rd_tune_set(name.encode(), int(value)), "tune_set")
    _products = None


def tune_get(name: str) -> int:
    v = C.c_int(0)
    check(load().rd_tune_get(name.encode(), C.cast(C.byref(v), C.c_void_p)), "tune_get")
    return v.value


# ---- profiler -----------------------------------------------------------------------------------
_prof_level = 0


def prof_enable(level):
    """0/False off; 1 = MFMA (roofline) kernel classes only; 2/True = every kernel class."""
    global _prof_level
    if level is True:
        level = 2
    check(load().rd_prof_enable(int(level)))
    _prof_level = int(level)


def prof_level_py() -> int:
    return _prof_level


def prof_reset():
    check(load().rd_prof_reset())


def prof_collect():
    arr = (ProfEntry * 64)()
    n = load().rd_prof_collect(C.cast(arr, C.c_void_p), 64)
    if n < 0:
        raise RuntimeError("rd_prof_collect failed")
    return [{"name": arr[i].name.decode(), "launches": arr[i].launches, "ms": arr[i].ms, "flops": arr[i].flops,
             "bytes": arr[i].bytes} for i in range(n)]
