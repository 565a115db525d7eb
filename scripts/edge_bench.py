"""Timing of the first / last convolution kernels at cfg-S shapes (N=32, 256x256, C=64), HIP events over 20 calls.
    python scripts/edge_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from resdepth_amd import ops

dev = "cuda:0"
n, t, c = 32, 256, 64
x = torch.randn(n, 3, t, t, device=dev)
w = torch.randn(c, 3, 3, 3, device=dev) * 0.1
s = torch.randn(n, t, t, c, device=dev)
dz = torch.randn(n, t, t, c, device=dev)
dout = torch.randn(n, 1, t, t, device=dev)
wl = torch.randn(1, c, 3, 3, device=dev) * 0.1
rm, rv, nbt = torch.zeros(c, device=dev), torch.ones(c, device=dev), torch.zeros((), device=dev, dtype=torch.long)
big = torch.empty(1 << 28, device=dev)          # 1 GiB scratch written between calls: evicts L2 / MALL


def timed(name, fn, bytes_, reps=20, flush=True):
    fn(); torch.cuda.synchronize()
    tot = 0.0
    for _ in range(reps):
        if flush:
            big.fill_(1.0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        tot += a.elapsed_time(b)
    ms = tot / reps
    print(f"{name:28s} {ms * 1e3:8.1f} us  {bytes_ / ms / 1e9:7.2f} TB/s (algorithmic){'  [cold]' if flush else '  [warm]'}")


P = n * t * t
for flush in (False, True):
    timed("first fwd", lambda: ops.conv3x3_first_fwd(x, w), 4.0 * P * (3 + c), flush=flush)
    timed("first fwd + BN stats", lambda: ops.conv3x3_first_fwd_bn(x, w, rm, rv, nbt), 4.0 * P * (3 + c), flush=flush)
    timed("first wgrad", lambda: ops.conv3x3_first_bwd_weight(x, dz), 4.0 * P * (3 + c), flush=flush)
    timed("last fwd", lambda: ops.conv3x3_last_fwd(s, wl, None, x), 4.0 * P * (c + 2), flush=flush)
    timed("last dgrad", lambda: ops.conv3x3_last_bwd_data(dout, wl, c), 4.0 * P * (c + 1), flush=flush)
    timed("last wgrad", lambda: ops.conv3x3_last_bwd_weight(s, dout), 4.0 * P * (c + 1), flush=flush)
