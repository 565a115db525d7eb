"""What do the fused epilogues of the 3x3 convolution kernels cost?  Per layer at the cfg-S shapes (N = 32): the forward with and
without the BatchNorm-statistics epilogue, the data gradient with and without the BN-backward statistics hook -- kernel time of the
MFMA kernel alone (rd_prof classes with a FLOP count), interleaved A / B, median of 7."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from resdepth_amd import ops, _lib

N = int(os.environ.get("BL_N", "32"))
dev = "cuda:0"
layers = [("enc1", 128, 64, 128), ("enc2", 64, 128, 256), ("enc3", 32, 256, 512), ("enc4", 16, 512, 512),
          ("dec0", 16, 512, 512), ("dec1", 32, 512, 256), ("dec2", 64, 256, 128), ("dec3", 128, 128, 64)]


def kernel_ms(fn):
    _lib.prof_reset(); _lib.prof_enable(2)
    fn()
    _lib.prof_enable(False)
    r = [e for e in _lib.prof_collect() if e["flops"] > 0]
    return sum(e["ms"] for e in r), sum(e["flops"] for e in r)


def ab(fa, fb, reps=7):
    fa(); fb(); torch.cuda.synchronize()
    a, b, fl = [], [], 0
    for _ in range(reps):
        m, fl = kernel_ms(fa); a.append(m)
        m, _ = kernel_ms(fb); b.append(m)
    a.sort(); b.sort()
    return a[len(a) // 2], b[len(b) // 2], fl


EPI = int(os.environ.get("NT_EPI", "-1"))       # 0: the LDS-staged epilogue of r03; -1: register-direct (default since r04)
_lib.tune_set("nt_epi", EPI)
print(f"nt_epi = {EPI}")
tot = [0.0] * 4
for name, h, cin, cout in layers:
    x = torch.randn(N, h, h, cin, device=dev)
    dz = torch.randn(N, h, h, cout, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    wf, wd = ops.pack_conv3x3_weight(w)
    # the data gradient's hook belongs to the block that PRODUCED x: its z has x's shape
    zin = torch.randn(N, h, h, cin, device=dev)
    mean, invstd = torch.zeros(cin, device=dev), torch.ones(cin, device=dev)
    gamma, beta = torch.ones(cin, device=dev), torch.zeros(cin, device=dev)
    hook = ops.BnHook(zin, mean, invstd, gamma, beta, 0.0, None, 1)
    f0, f1, fl = ab(lambda: ops.conv3x3_fwd(x, wf), lambda: ops.conv3x3_fwd_stats(x, wf))
    d0, d1, _ = ab(lambda: ops.conv3x3_bwd_data(dz, wd), lambda: ops.conv3x3_bwd_data(dz, wd, hook))
    for i, v in enumerate((f0, f1, d0, d1)):
        tot[i] += v
    print(f"{name:5s} H={h:3d} Cin={cin:3d} Cout={cout:3d} | fwd {f0:.4f} ms ({fl / f0 / 1e9:6.1f} TF)  +stats {f1:.4f} ms ({(f1 / f0 - 1) * 100:+5.1f} %)"
          f" | dgrad {d0:.4f} ms ({fl / d0 / 1e9:6.1f} TF)  +hook {d1:.4f} ms ({(d1 / d0 - 1) * 100:+5.1f} %)")
print(f"totals: fwd {tot[0]:.3f} -> {tot[1]:.3f} ms with statistics; dgrad {tot[2]:.3f} -> {tot[3]:.3f} ms with the hook")
