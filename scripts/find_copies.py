"""Which torch-level copies / fills does one training step issue?  (torch.profiler on one step; r05: 42 CPU-side scalar conversions from
the optimizer's per-parameter step counters, one fill from autograd's ones_like -- no device copy of ours; the ~6 small blit kernels per
step that rocprofv3 --stats lists are not torch operations.)"""
import sys, torch, importlib.util
sys.path.insert(0, ".")
spec = importlib.util.spec_from_file_location("bench_mod", "bench.py"); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
from resdepth_amd import _lib; _lib.load()
dev = torch.device("cuda", 0)
tb = b.TrainBench(b.WORKLOADS["S"], 32, dev); tb.attach_optimizer()
for _ in range(3): tb.step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    tb.step(); torch.cuda.synchronize()
import collections
cnt = collections.Counter()
for e in prof.events():
    n = e.name
    if n in ("aten::_to_copy", "aten::clone", "aten::fill_", "aten::zero_", "aten::copy_") or "Memcpy" in n or "Memset" in n:
        st = [s_ for s_ in (e.stack or []) if "site-packages/torch" not in s_ and "dist-packages/torch" not in s_ and "<built-in" not in s_]
        cnt[(n, tuple(st[:3]), str(getattr(e, "input_shapes", "")))] += 1
for (n, st, sh), c in cnt.most_common(30):
    print(c, n, sh, " <- ".join(x[-70:] for x in st))
