"""Time the cfg-S bottleneck convolution (8 x 8 x 512 -> 512 at batch 32) forward / data gradient: generic vs split-K patch kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from resdepth_amd import ops, _lib
_lib.ensure_splitk_workspace("cuda:0")
x = torch.randn(32, 8, 8, 512, device="cuda:0"); w = torch.randn(512, 512, 3, 3, device="cuda:0") * 0.05
wf, wd = ops.pack_conv3x3_weight(w)
for knob in (0, -1):
    _lib.tune_set("nt_splitk", knob)
    for _ in range(3): ops.conv3x3_fwd(x, wf); ops.conv3x3_bwd_data(x, wd)
    torch.cuda.synchronize()
    _lib.prof_reset(); _lib.prof_enable(2)
    for _ in range(20): ops.conv3x3_fwd(x, wf); ops.conv3x3_bwd_data(x, wd)
    _lib.prof_enable(False)
    for e in _lib.prof_collect():
        print("nt_splitk", knob, e["name"], round(e["ms"] / e["launches"], 4), "ms", round(e["flops"] / e["ms"] / 1e9, 1), "TF")
_lib.tune_set("nt_splitk", -1)
