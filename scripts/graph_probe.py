"""The captured training iteration against the eager one: bit comparison of weights / BN buffers / moments / losses over K steps
(learning rate changed on the way), host enqueue time and wall time per step of both.
    python scripts/graph_probe.py [batch] [steps]"""
import os, sys, time, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from resdepth_amd import UNet, FusedAdam, synthetic_batch
from resdepth_amd.graph import GraphedTrainStep

dev = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
K = int(sys.argv[2]) if len(sys.argv) > 2 else 12
kw = dict(n_input_channels=3, start_kernel=64, depth=5, bias_conv_layer=True)
torch.manual_seed(0)
m0 = UNet(**kw)
sd0 = copy.deepcopy(m0.state_dict())
batches = []
for i in range(4):
    b = synthetic_batch(n, 3, 256, seed=5 + i)
    batches.append((b["input"].to(dev), b["target"].to(dev), b["loss_mask"].to(dev), b["dsm_mean"].to(torch.float32).to(dev), b["dsm_std"].to(dev)))


def run(graphed):
    model = UNet(**kw)
    model.load_state_dict(sd0)
    model = model.to(dev).train()
    opt = FusedAdam(model.parameters(), lr=2e-4, weight_decay=1e-5)
    st = GraphedTrainStep(model, opt, warmup=2 if graphed else 1 << 60)
    losses = []
    for k in range(K):
        if k == K // 2:
            opt.param_groups[0]["lr"] = 1e-4
        losses.append(st(*batches[k % 4]).clone())
    torch.cuda.synchronize()
    return model, opt, st, torch.stack(losses).cpu()


me, oe, se, le = run(False)
mg, og, sg, lg = run(True)
print(f"batch {n}, {K} steps: replays {sg.replays}, last eager reason: {sg.why_eager}")
print("losses equal:", torch.equal(le, lg), le[-3:].tolist(), lg[-3:].tolist())
bad = [k for k, v in me.state_dict().items() if not torch.equal(v, mg.state_dict()[k])]
print("state_dict entries differing:", len(bad), bad[:3])
sde, sdg = oe.state_dict(), og.state_dict()
badm = [(i, k) for i in sde["state"] for k in sde["state"][i] if not torch.equal(sde["state"][i][k].cpu(), sdg["state"][i][k].cpu())]
print("optimizer state entries differing:", len(badm), badm[:3], "| step", float(sdg["state"][0]["step"]))
for name, st in (("eager", se), ("graph", sg)):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(20):
        st(*batches[k % 4])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name}: host {(t1 - t0) / 20 * 1e3:.3f} ms/step, wall {(t2 - t0) / 20 * 1e3:.3f} ms/step -> {n * 20 / (t2 - t0):.1f} tiles/s")
