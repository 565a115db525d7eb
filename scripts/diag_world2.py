"""Diagnosis: world-2 (4+4) vs one process at 8 tiles -- is the forward bit-identical, which gradients differ by how much."""
import os, sys, pathlib, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_dp_world2_gpu as T
world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
sync = int(sys.argv[2]) if len(sys.argv) > 2 else 1
tmp = pathlib.Path(tempfile.mkdtemp())
pin = sys.argv[3] if len(sys.argv) > 3 else ""
outs = T.run_world(tmp, "train", world=world, coll="staged", sync_bn=sync, batch=8, steps=1, **({"tune": pin} if pin else {}))
ref = T._single_process(8, 1, tune=pin)
y = torch.cat([o["y0"] for o in outs])
d = (y - ref["y0"]).abs()
print("forward: max |dy|", float(d.max()), "differing elements", int((d != 0).sum()), "of", d.numel())
print("loss", outs[0]["losses"], ref["losses"])
for k, v in ref["bufs0"].items():
    if v.dtype.is_floating_point:
        e = T.rel_l2(outs[0]["bufs0"][k], v)
        if e > 0: print("buf", k, e)
for k, g in ref["grads0"].items():
    print(f"{k:32s} {T.rel_l2(outs[0]['grads0'][k], g):.3e}")
