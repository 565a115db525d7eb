#!/usr/bin/env python3
"""Why is the from-rasters step slower than the resident-batch step?  Three timings on one box:
  (a) the bench's resident randn batch;  (b) ONE sampler batch kept resident (same operand statistics as (c), no sampling work);
  (c) a fresh sampler batch every step (stream_batches, prefetch 1)."""
import importlib.util, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
from resdepth_amd import _lib
_lib.load()
dev = torch.device("cuda", 0)
wl = b.WORKLOADS["S"]
for rnd in range(2):
    tb = b.TrainBench(wl, 32, dev); tb.attach_optimizer()
    dt, ev = tb.timed(15, 4); print("(a) resident randn batch      ms/step %.3f" % b._median(ev))
    tr = b.TrainBench(wl, 32, dev, from_rasters=True); tr.attach_optimizer()
    bb = next(tr.batch_iter)
    tb2 = b.TrainBench(wl, 32, dev); tb2.attach_optimizer()
    tb2.x, tb2.y, tb2.mask, tb2.mean, tb2.std = bb["input"], bb["target"], bb["loss_mask"], bb["dsm_mean"], bb["dsm_std"]
    dt, ev = tb2.timed(15, 4); print("(b) resident SAMPLER batch    ms/step %.3f" % b._median(ev))
    dt, ev = tr.timed(15, 4); print("(c) fresh sampler batch/step  ms/step %.3f" % b._median(ev))
    del tb, tb2, tr
