import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from resdepth_amd import UNet, FusedAdam, synthetic_batch
from resdepth_amd.plan import PlannedTrainStep
N, variant = int(sys.argv[1]), sys.argv[2]
dev = "cuda:0"
torch.manual_seed(0)
model = UNet(n_input_channels=3, start_kernel=64, depth=5, bias_conv_layer=True).to(dev).train()
opt = FusedAdam(model.parameters(), lr=2e-4, weight_decay=1e-5)
b = synthetic_batch(N, 3, 256, seed=1234)
batch = (b["input"].to(dev), b["target"].to(dev), b["loss_mask"].to(dev), b["dsm_mean"].float().to(dev), b["dsm_std"].float().to(dev))
step = PlannedTrainStep(model, opt, warmup=2, verify=False)
flags, losses = [], []
for k in range(7):
    l = step(*batch)
    losses.append(l.clone())
    flags.append(torch.isfinite(model._flat_param).all())
    if variant == "sync3" and k == 2:
        torch.cuda.synchronize()
    if variant == "sync4" and k == 3:
        torch.cuda.synchronize()
torch.cuda.synchronize()
print(variant, [bool(f) for f in flags], [round(float(x), 4) for x in losses])
