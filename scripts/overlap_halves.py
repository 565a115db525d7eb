"""Bounding experiment: would pipelining two half batches through the forward pass (so that one half's HBM-bound BN / pool passes
run under the other half's MFMA-bound convolutions) pay?  Upper bound: two INDEPENDENT forwards at batch 16 on two streams (no
BatchNorm coupling between the halves at all) against one forward at batch 32.  Training-mode forward, cfg-S architecture."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from resdepth_amd import UNet, synthetic_batch

dev = "cuda:0"
torch.manual_seed(0)
kw = dict(n_input_channels=3, start_kernel=64, depth=5, bias_conv_layer=True)
m32 = UNet(**kw).to(dev).train()
ma, mb = UNet(**kw).to(dev).train(), UNet(**kw).to(dev).train()
x = synthetic_batch(32, 3, 256)["input"].to(dev)
xa, xb = x[:16].contiguous(), x[16:].contiguous()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def one():
    with torch.no_grad():
        m32(x)


def two():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.no_grad():
        with torch.cuda.stream(s1):
            ma(xa)
        with torch.cuda.stream(s2):
            mb(xb)
    cur.wait_stream(s1); cur.wait_stream(s2)


def seq16():
    with torch.no_grad():
        ma(xa); mb(xb)


def timed(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for r in range(3):
    print(f"round {r}: forward batch 32: {timed(one):.3f} ms | two batch-16 forwards, one stream: {timed(seq16):.3f} ms | on two streams: {timed(two):.3f} ms")
