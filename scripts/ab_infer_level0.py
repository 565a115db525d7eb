"""Interleaved A/B of the cfg-G sweep with level 0 of the inference forward as one kernel (UNet.fused_first_eval) vs the r03 route."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils.data import DataLoader
from resdepth_amd import UNet, SyntheticRasterTiles, predict_linear_blend

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = UNet(n_input_channels=3, start_kernel=64, depth=5, bias_conv_layer=True).to(dev).eval()
R = int(os.environ.get("RASTER", "4096"))
ds = SyntheticRasterTiles(R, R, 3, tile_size=256, seed=1)
batches = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()} for b in DataLoader(ds, batch_size=32, shuffle=False)]


class Loader(list):
    dataset = ds


loader = Loader(batches)
sums = {}
for rnd in range(3):
    for fused in (False, True):
        model.fused_first_eval = fused
        predict_linear_blend(loader, model)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            out = predict_linear_blend(loader, model)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 2
        sums[fused] = float(out.sum())
        print(f"round {rnd} fused_first_eval={fused}: {len(ds) / dt:9.1f} tiles/s  ({dt * 1e3:.1f} ms per sweep)  checksum {sums[fused]!r}")
assert sums[False] == sums[True]
