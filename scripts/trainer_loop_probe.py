"""bench.trainer_loop_measurement a few times, eager and planned (a 1339 tiles/s outlier appeared once in the default bench run)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda:0")
wl = bench.WORKLOADS["S"]
for i in range(3):
    for plan in (False, True):
        r = bench.trainer_loop_measurement(dev, wl, 32, prefetch=1, launch_plan=plan)
        print("round", i, "plan" if plan else "eager", r["tiles_per_s"], r["ms_per_iteration"], r["loss_avg"], flush=True)
