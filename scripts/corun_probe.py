"""How long does a SMALL main-stream kernel take beside the weight-gradient stream's strip kernel?  (The production timeline shows
the BN-backward statistics finalize at 117-138 us beside wgrad_strip_tr, 6-20 us alone.)  HIP-event time of the small kernel
alone / issued while a strip launch (or a convt_wgrad launch) is running on a second stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from resdepth_amd import ops, _lib

dev = "cuda:0"
torch.manual_seed(0)
x = ops.amax_of(torch.randn(32, 128, 128, 64, device=dev))
dz = ops.amax_of(torch.randn(32, 128, 128, 128, device=dev))
xt = ops.amax_of(torch.randn(32, 64, 64, 128, device=dev))
dot = ops.amax_of(torch.randn(32, 128, 128, 128, device=dev))
part = torch.randn(8192, 4 * 64, device=dev)
big = torch.empty(4 << 20, device=dev)                      # 16 MB
side = torch.cuda.Stream()


def small_fin2():
    _lib.tune_set("bn_fin2", 1); ops.bn_bwd_stats_finalize([(part, 8192)], 64)


def small_fin1():
    _lib.tune_set("bn_fin2", 0); ops.bn_bwd_stats_finalize([(part, 8192)], 64); _lib.tune_set("bn_fin2", 1)


def small_zero():
    _lib.zero_(big)


def small_copy():
    big.copy_(part.view(-1)[: big.numel()] if part.numel() >= big.numel() else big)


def heavy_strip():
    ops.conv3x3_bwd_weight(x, dz)


def heavy_convt():
    ops.convt2x2_bwd_weight(xt, dot)


def timed(small, heavy, reps=20, delay_us=60):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        if heavy is not None:
            with torch.cuda.stream(side):
                heavy()
            torch.cuda._sleep(int(delay_us * 2100))          # main stream: let the heavy kernel get going
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); small(); b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b) * 1e3)
    out.sort()
    return out[len(out) // 2], out[0], out[-1]


for name, small in (("finalize 2-stage", small_fin2), ("finalize 1-stage", small_fin1), ("rd_zero 16 MB", small_zero)):
    for hname, heavy in (("alone", None), ("beside wgrad_strip", heavy_strip), ("beside convt_wgrad", heavy_convt)):
        med, lo, hi = timed(small, heavy)
        print(f"{name:18s} {hname:20s} median {med:7.1f} us  min {lo:7.1f}  max {hi:7.1f}")

# ---- the production sequence: data gradient on the main stream, the strip kernel on the side stream beside it, THEN the finalize
print("--- finalize right behind a data-gradient launch that co-ran with the strip kernel")
_, wd = ops.pack_conv3x3_weight(torch.randn(128, 64, 3, 3, device=dev) * 0.05)
for name, small in (("finalize 2-stage", small_fin2), ("finalize 1-stage", small_fin1), ("rd_zero 16 MB", small_zero)):
    for order in ("dgrad first", "strip first"):
        ts = []
        for _ in range(12):
            torch.cuda.synchronize()
            if order == "strip first":
                with torch.cuda.stream(side):
                    heavy_strip()
                ops.conv3x3_bwd_data(dz, wd)
            else:
                ops.conv3x3_bwd_data(dz, wd)
                with torch.cuda.stream(side):
                    heavy_strip()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); small(); b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        ts.sort()
        print(f"{name:18s} {order:12s} median {ts[len(ts) // 2]:7.1f} us  min {ts[0]:7.1f}  max {ts[-1]:7.1f}")

print("--- the same with a HIGH-priority side stream (dgrad first), and with the small kernel on a third stream")
hi = torch.cuda.Stream(priority=-1)
third = torch.cuda.Stream()
for name, small in (("finalize 2-stage", small_fin2), ("rd_zero 16 MB", small_zero)):
    for variant in ("side high-prio", "small on 3rd stream", "side high-prio, total"):
        ts = []
        for _ in range(12):
            torch.cuda.synchronize()
            t0 = torch.cuda.Event(enable_timing=True); t0.record()
            ops.conv3x3_bwd_data(dz, wd)
            with torch.cuda.stream(hi if variant.startswith("side high") else side):
                heavy_strip()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if variant == "small on 3rd stream":
                with torch.cuda.stream(third):
                    torch.cuda._sleep(int(250 * 2100))
                    a.record(); small(); b.record()
            else:
                a.record(); small(); b.record()
            torch.cuda.synchronize()
            t1 = torch.cuda.Event(enable_timing=True); t1.record(); torch.cuda.synchronize()
            ts.append((t0.elapsed_time(t1) if variant.endswith("total") else a.elapsed_time(b)) * 1e3)
        ts.sort()
        print(f"{name:18s} {variant:24s} median {ts[len(ts) // 2]:7.1f} us  min {ts[0]:7.1f}  max {ts[-1]:7.1f}")
# totals of the two orders with the normal side stream
for order in ("dgrad first", "strip first"):
    ts = []
    for _ in range(12):
        torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True); t0.record()
        if order == "strip first":
            with torch.cuda.stream(side):
                heavy_strip()
            ops.conv3x3_bwd_data(dz, wd)
        else:
            ops.conv3x3_bwd_data(dz, wd)
            with torch.cuda.stream(side):
                heavy_strip()
        small_fin2()
        torch.cuda.synchronize()
        t1 = torch.cuda.Event(enable_timing=True); t1.record(); torch.cuda.synchronize()
        ts.append(t0.elapsed_time(t1) * 1e3)
    ts.sort()
    print(f"total (dgrad + strip + finalize), {order}: median {ts[len(ts) // 2]:7.1f} us")
