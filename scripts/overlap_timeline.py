#!/usr/bin/env python3
"""Who runs when: classify every kernel of a `rocprofv3 --kernel-trace` CSV of the PRODUCTION step (two streams) as
MFMA-class (3x3 / transposed convolutions, generic GEMMs) or HBM-class (everything else) and integrate the timeline of the
last `--steps` steps: time with only MFMA-class kernels running, only HBM-class, both, none.  The "only HBM" + "none" part is
what better overlap could still hide under the MFMA-bound work.
    python scripts/overlap_timeline.py gpurun_out/tl/bench_kernel_trace.csv --steps 5"""
import argparse
import csv
import re
import sys

MFMA = re.compile(r"conv3_halo|wgrad_strip|convt_|igemm_nt|wgrad_tn|conv3_first|gemm")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--marker", default="adam", help="kernel-name substring that ends a step")
    a = ap.parse_args()
    rows = []
    with open(a.csv) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    ends = [e for s, e, n in rows if a.marker in n.lower()]
    if len(ends) < a.steps + 1:
        sys.exit(f"only {len(ends)} step markers")
    t0, t1 = ends[-a.steps - 1], ends[-1]
    ev = []
    per = {}
    for s, e, n in rows:
        if e <= t0 or s >= t1:
            continue
        s, e = max(s, t0), min(e, t1)
        cls = 0 if MFMA.search(n) else 1
        ev.append((s, 1, cls))
        ev.append((e, -1, cls))
        short = re.sub(r"\(.*", "", n).replace("void rd::", "")
        per.setdefault(short, [0, 0.0])
        per[short][0] += 1
        per[short][1] += (e - s)
    ev.sort()
    active = [0, 0]
    acc = {"mfma_only": 0, "hbm_only": 0, "both": 0, "idle": 0}
    hbm_alone_by = {}
    last = t0
    for t, d, cls in ev:
        dt = t - last
        if dt > 0:
            key = "both" if active[0] and active[1] else "mfma_only" if active[0] else "hbm_only" if active[1] else "idle"
            acc[key] += dt
        active[cls] += d
        last = t
    acc["idle"] += t1 - last
    tot = (t1 - t0) / a.steps / 1e6
    print(f"step {tot:.3f} ms over {a.steps} steps")
    for k, v in acc.items():
        print(f"  {k:10s} {v / a.steps / 1e6:7.3f} ms  {100.0 * v / (t1 - t0):5.1f} %")
    print("kernel time per step (sum of durations, ms):")
    for n, (c, d) in sorted(per.items(), key=lambda x: -x[1][1])[:24]:
        print(f"  {d / a.steps / 1e6:7.3f}  x{c / a.steps:5.1f}  {'M' if MFMA.search(n) else 'h'}  {n[:90]}")


if __name__ == "__main__":
    main()
