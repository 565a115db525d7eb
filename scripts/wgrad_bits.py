"""Weight gradients of the cfg-S shapes (+ a ragged batch and the 8 x 8 level) -> one .pt file; run once per library build
(RESDEPTH_HIP_LIB) and compare the files: the rotating-register strip kernel must reproduce the ring kernel bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from resdepth_amd import ops

torch.manual_seed(7)
out = {}
for name, n, h, cin, cout in [("enc1", 32, 128, 64, 128), ("enc3", 32, 32, 256, 512), ("bott", 32, 8, 512, 512), ("bott_odd", 5, 8, 512, 512),
                              ("dec1", 32, 32, 512, 256), ("dec3", 32, 128, 128, 64), ("rag", 3, 48, 64, 64), ("enc4", 7, 16, 512, 512)]:
    x = ops.amax_of(torch.randn(n, h, h, cin, device="cuda"))
    dz = ops.amax_of(torch.randn(n, h, h, cout, device="cuda") * 3e-3)
    out[name] = ops.conv3x3_bwd_weight(x, dz).cpu()
    ref = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2).double().cpu(), (cout, cin, 3, 3), dz.permute(0, 3, 1, 2).double().cpu(), padding=1) if n * h * h * cin * cout < 3e10 else None
    if ref is not None:
        err = (out[name].double() - ref).abs().max() / ref.abs().max()
        print(name, "rel err vs fp64", float(err))
torch.save(out, sys.argv[1])
if len(sys.argv) > 2:
    other = torch.load(sys.argv[2])
    for k in out:
        print(k, "bit-identical" if torch.equal(out[k], other[k]) else "DIFFERENT %g" % float((out[k] - other[k]).abs().max()))
