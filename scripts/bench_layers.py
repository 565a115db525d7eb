"""Per-layer timing of the MFMA kernels at the cfg-S shapes (N=32).  Tuning aid; RD_TUNE="nt_tile=0,tn_blocks=512,..."
overrides the tile / split heuristics (include/resdepth_hip.h: rd_tune_set)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from resdepth_amd import ops, _lib

N = int(os.environ.get("BL_N", "32"))
dev = "cuda:0"
layers = [  # name, H, Cin, Cout
    ("enc1", 128, 64, 128), ("enc2", 64, 128, 256), ("enc3", 32, 256, 512), ("enc4", 16, 512, 512),
    ("bott", 8, 512, 512), ("dec0", 16, 512, 512), ("dec1", 32, 512, 256), ("dec2", 64, 256, 128), ("dec3", 128, 128, 64)]
convt = [("up0", 8, 512), ("up1", 16, 512), ("up2", 32, 256), ("up3", 64, 128), ("up4", 128, 64)]
which = os.environ.get("BL_WHICH", "fwd,dgrad,wgrad,convt").split(",")
reps = 5
TAG = os.environ.get("BL_TAG", "1") == "1"      # BL_TAG=0: untagged operands -> six-product fallback bodies


def timed(fn):
    fn(); torch.cuda.synchronize()
    _lib.prof_reset(); _lib.prof_enable(2)
    for _ in range(reps):
        fn()
    _lib.prof_enable(False)
    r = [e for e in _lib.prof_collect() if e["flops"] > 0]
    ms = sum(e["ms"] for e in r) / reps
    fl = sum(e["flops"] for e in r) / reps
    return ms, fl / (ms * 1e-3) / 1e12


tot = {}
print(f"cfg RD_TUNE={os.environ.get('RD_TUNE')} N={N}")
for name, h, cin, cout in layers:
    x = torch.randn(N, h, h, cin, device=dev)
    dz = torch.randn(N, h, h, cout, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    if os.environ.get("BL_DATA") == "zero":          # power experiment: no operand toggling in the matrix pipe
        x.zero_(); dz.zero_(); w.zero_()
    elif os.environ.get("BL_DATA") == "one":         # exact bf16 values: the mid / lo split terms are all zero
        x.fill_(1.0); dz.fill_(1.0); w.fill_(0.5)
    wf, wd = ops.pack_conv3x3_weight(w)
    if TAG:
        x, dz = ops.amax_of(x), ops.amax_of(dz)       # magnitude slots: the launches take the three-product bodies
    row = f"{name:5s} M={N*h*h:7d} Cin={cin:3d} Cout={cout:3d} "
    if "fwd" in which:
        ms, tf = timed(lambda: ops.conv3x3_fwd(x, wf)); row += f"| fwd {ms:6.3f} ms {tf:6.1f} TF "; tot["fwd"] = tot.get("fwd", 0) + ms
    if "dgrad" in which:
        ms, tf = timed(lambda: ops.conv3x3_bwd_data(dz, wd)); row += f"| dgrad {ms:6.3f} ms {tf:6.1f} TF "; tot["dgrad"] = tot.get("dgrad", 0) + ms
    if "wgrad" in which:
        ms, tf = timed(lambda: ops.conv3x3_bwd_weight(x, dz)); row += f"| wgrad {ms:6.3f} ms {tf:6.1f} TF "; tot["wgrad"] = tot.get("wgrad", 0) + ms
    print(row)
if "convt" in which:
    for name, h, c in convt:
        x = torch.randn(N, h, h, c, device=dev)
        do = torch.randn(N, 2 * h, 2 * h, c, device=dev)
        skip = torch.randn(N, 2 * h, 2 * h, c, device=dev)
        w = torch.randn(c, c, 2, 2, device=dev) * 0.05
        b = torch.randn(c, device=dev)
        if os.environ.get("BL_DATA") == "zero":
            x.zero_(); do.zero_(); skip.zero_(); w.zero_(); b.zero_()
        wtf, wtd = ops.pack_convt2x2_weight(w)
        if TAG:
            x, do = ops.amax_of(x), ops.amax_of(do)
        row = f"{name:5s} M={N*h*h:7d} C={c:3d}          "
        ms, tf = timed(lambda: ops.convt2x2_fwd(x, wtf, b, skip)); row += f"| fwd {ms:6.3f} ms {tf:6.1f} TF "; tot["tfwd"] = tot.get("tfwd", 0) + ms
        ms, tf = timed(lambda: ops.convt2x2_bwd_data(do, wtd)); row += f"| dgrad {ms:6.3f} ms {tf:6.1f} TF "; tot["tdgrad"] = tot.get("tdgrad", 0) + ms
        ms, tf = timed(lambda: ops.convt2x2_bwd_weight(x, do)); row += f"| wgrad {ms:6.3f} ms {tf:6.1f} TF "; tot["twgrad"] = tot.get("twgrad", 0) + ms
        print(row)
print("totals ms:", {k: round(v, 3) for k, v in tot.items()})
