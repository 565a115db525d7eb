#!/usr/bin/env python3
"""Per-launch durations (us) of the small reduction / finalize kernels in the last step of a rocprofv3 --kernel-trace CSV:
    python scripts/small_kernels.py gpurun_out/tl/bench_kernel_trace.csv [steps-in-trace]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 11
d = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if any(k in n for k in ("finalize", "slab_reduce", "reduce_partials", "partial_reduce", "l1_", "pack_")):
        d[n[:64]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in sorted(d.items()):
    per = max(1, len(v) // steps)
    print(f"{n:64s} x{per:3d}  sum {sum(v[-per:]):6.0f} us : " + " ".join(f"{x:.0f}" for x in v[-per:]))
