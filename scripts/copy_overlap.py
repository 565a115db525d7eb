#!/usr/bin/env python3
"""How much of the traced memory copies of a run lies UNDER running kernels?
    python scripts/copy_overlap.py <rocprofv3 output dir> [H2D|D2H] [min_us]
Reads *kernel_trace.csv and *memory_copy_trace.csv (rocprofv3 --kernel-trace --memory-copy-trace), drops the first third of
the kernels (warm-up) and reports, for copies of the given direction longer than min_us: total time, time under kernels, exposed."""
import csv, glob, sys

d = sys.argv[1]
want = (sys.argv[2] if len(sys.argv) > 2 else "D2H").upper()
min_ns = float(sys.argv[3]) * 1e3 if len(sys.argv) > 3 else 50e3
kf = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
mf = glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True)
if not kf or not mf:
    sys.exit("no kernel / memory-copy trace under " + d)
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(kf[0])))
rows = rows[len(rows) // 3:]
# copies to / from page-locked host memory may run as a BLIT KERNEL (__amd_rocclr_copyBuffer) instead of an SDMA transfer: those
# show up in the kernel trace, not in the memory-copy trace
blit = [(s, e) for s, e, n in rows if "copyBuffer" in n and e - s >= min_ns]
ev = [(s, e) for s, e, n in rows if "copyBuffer" not in n]
iv, (cs, ce) = [], ev[0]
for s, e in ev[1:]:
    if s > ce:
        iv.append((cs, ce)); cs, ce = s, e
    else:
        ce = max(ce, e)
iv.append((cs, ce))
key = "HOST_TO_DEVICE" if want == "H2D" else "DEVICE_TO_HOST"
cp = []
for r in csv.DictReader(open(mf[0])):
    kind = (r.get("Direction") or r.get("Kind") or "").upper()
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if key in kind and s >= ev[0][0] and e - s >= min_ns:
        cp.append((s, e))
if not cp and blit:
    print(f"(no {want} record in the memory-copy trace: using the {len(blit)} blit kernels >= {min_ns / 1e3:.0f} us of the kernel trace)")
    cp = blit
cp.sort()
tot = sum(e - s for s, e in cp)
under = 0
for s, e in cp:
    for a, b in iv:
        if b <= s:
            continue
        if a >= e:
            break
        under += max(0, min(e, b) - max(s, a))
print(f"{want} copies >= {min_ns / 1e3:.0f} us: n {len(cp)}  total {tot / 1e6:.2f} ms  under kernels {under / 1e6:.2f} ms "
      f"({100 * under / max(tot, 1):.1f} %)  exposed {(tot - under) / 1e6:.2f} ms  kernel span {(ev[-1][1] - ev[0][0]) / 1e6:.1f} ms")
if cp:
    print(f"last kernel end -> last copy end: {(cp[-1][1] - ev[-1][1]) / 1e6:.3f} ms")
