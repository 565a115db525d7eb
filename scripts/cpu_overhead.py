"""How long does the host need to ENQUEUE one training step (python + ctypes + torch allocator) vs the GPU to run it?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from resdepth_amd import UNet, FusedAdam, masked_l1_loss

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = UNet(n_input_channels=3, start_kernel=64, depth=5).to(dev).train()
opt = FusedAdam(model.parameters(), lr=2e-4, weight_decay=1e-5)
n = 32
x = torch.randn(n, 3, 256, 256, device=dev); y = torch.randn(n, 1, 256, 256, device=dev)
mask = torch.rand(n, 1, 256, 256, device=dev) > 0.05
mean = torch.zeros(n, dtype=torch.float64); std = torch.ones(n)
if os.environ.get('DEV_STATS', '1') == '1':
    mean, std = mean.to(torch.float32).to(dev), std.to(dev)

def step():
    for p in model.parameters():
        p.grad = None
    loss = masked_l1_loss(model(x), y, mask, mean, std)
    loss.backward()
    opt.step()

for _ in range(3):
    step()
torch.cuda.synchronize()
for K in (1, 1, 2, 10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"K={K}: enqueue {1e3*(t1-t0)/K:.2f} ms/step   total {1e3*(t2-t0)/K:.2f} ms/step")
import cProfile, pstats
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable(); step(); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
