"""Where does the data-parallel machinery cost time at world size 1?  torchrun --nproc-per-node 1 scripts/dp_overhead.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from resdepth_amd import UNet, FusedAdam, masked_l1_loss, dp

dist.init_process_group("nccl", rank=0, world_size=1, init_method="tcp://127.0.0.1:29519", **({"device_id": torch.device("cuda:0")} if os.environ.get("DEVID") else {}))
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
n = 32
x = torch.randn(n, 3, 256, 256, device=dev); y = torch.randn(n, 1, 256, 256, device=dev)
mask = torch.rand(n, 1, 256, 256, device=dev) > 0.05
mean = torch.zeros(n, device=dev); std = torch.ones(n, device=dev)


def run(tag, attach, bucket=16 << 20, loss_sync=True):
    torch.manual_seed(0)
    model = UNet(n_input_channels=3, start_kernel=64, depth=5).to(dev).train()
    opt = FusedAdam(model.parameters(), lr=2e-4, weight_decay=1e-5)
    gs = dp.attach(model, bucket_bytes=bucket) if attach else None

    def step():
        for p in model.parameters():
            p.grad = None
        loss = masked_l1_loss(model(x), y, mask, mean, std, grad_sync=gs if loss_sync else None)
        loss.backward()
        opt.step()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    print(f"{tag:40s} {1e3 * (time.perf_counter() - t0) / 10:.2f} ms/step", flush=True)


run("no DP", False)
run("DP, 16 MB buckets", True)
run("DP, one bucket at the end", True, bucket=1 << 40)
run("DP, 16 MB buckets, local loss normaliser", True, loss_sync=False)
run("DP, 4 MB buckets", True, bucket=4 << 20)
