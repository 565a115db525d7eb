"""Can an HBM-bound kernel hide under an MFMA-bound one on this part?  Times a conv3x3 forward (split-bf16 halo kernel)
and a BN-apply+pool pass back to back on one stream vs concurrently on two streams."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from resdepth_amd import ops

dev = "cuda:0"
N = 32
x = torch.randn(N, 64, 64, 256, device=dev); w = torch.randn(128, 256, 3, 3, device=dev) * 0.02
wf, _ = ops.pack_conv3x3_weight(w)
z = torch.randn(N, 256, 256, 64, device=dev)
mean = torch.zeros(64, device=dev); invstd = torch.ones(64, device=dev); gamma = torch.ones(64, device=dev); beta = torch.zeros(64, device=dev)
R = 10


def mfma():
    for _ in range(R):
        ops.conv3x3_fwd(x, wf)


def hbm():
    for _ in range(R // 2):
        ops.bn_act_pool_fwd(z, mean, invstd, gamma, beta, 0.0, True, want_a=False)


def timed(fn):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0)


s2 = torch.cuda.Stream()
def both():
    with torch.cuda.stream(s2):
        hbm()
    mfma()
    torch.cuda.current_stream().wait_stream(s2)

a, b = timed(mfma), timed(hbm)
c = timed(both)
print(f"mfma alone {a:.2f} ms, hbm alone {b:.2f} ms, serial sum {a+b:.2f} ms, concurrent {c:.2f} ms")
