"""Run one conv3x3 (fwd|dgrad|wgrad) or transposed-conv (tfwd|tdgrad|twgrad, C = CIN) layer a few times (profiling aid):
    python scripts/one_layer.py H CIN COUT [mode] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from resdepth_amd import ops

h, cin, cout = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
what = sys.argv[4] if len(sys.argv) > 4 else "fwd"
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
N = int(os.environ.get("BL_N", "32"))
x = torch.randn(N, h, h, cin, device="cuda:0")
dz = torch.randn(N, h, h, cout, device="cuda:0")
w = torch.randn(cout, cin, 3, 3, device="cuda:0") * 0.05
wf, wd = ops.pack_conv3x3_weight(w)
if what in ("tfwd", "tdgrad", "twgrad"):          # transposed convolution C -> C at H x H (coarse)
    c = cin
    wt = torch.randn(c, c, 2, 2, device="cuda:0") * 0.05
    wtf, wtd = ops.pack_convt2x2_weight(wt)
    xc = torch.randn(N, h, h, c, device="cuda:0")
    do = torch.randn(N, 2 * h, 2 * h, c, device="cuda:0")
    for _ in range(reps):
        if what == "tfwd":
            ops.convt2x2_fwd(xc, wtf, None, None)
        elif what == "tdgrad":
            ops.convt2x2_bwd_data(do, wtd)
        else:
            ops.convt2x2_bwd_weight(xc, do)
    torch.cuda.synchronize()
    sys.exit(0)
for _ in range(reps):
    if what == "fwd":
        ops.conv3x3_fwd(x, wf)
    elif what == "dgrad":
        ops.conv3x3_bwd_data(dz, wd)
    else:
        ops.conv3x3_bwd_weight(x, dz)
torch.cuda.synchronize()
