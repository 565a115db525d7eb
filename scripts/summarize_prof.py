#!/usr/bin/env python3
"""Condense rocprofv3 output (scripts/profile.sh -> gpurun_out/prof/) into profiles/r<NN>_*:
  r01_kernel_stats.csv      kernel-trace --stats summary of `bench.py` (per kernel symbol: calls, avg, total, %)
  r01_summary.json          per kernel symbol: avg duration, HBM bytes per launch from the separate --pmc passes
                            (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16-byte coalesced reads on
                            gfx950, WRITE_SIZE as reported; both in KiB units of the counter), L2 hit rate,
                            effective clock and MFMA pipe utilisation."""
import collections, csv, json, os, re, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
DST = os.path.join(ROOT, "profiles")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r01"


def sym(name):
    """rocprof kernel name -> the short symbol used by rd_prof / bench.py."""
    m = re.match(r"void rd::igemm_nt(_split)?_kernel<(\d+), (\d+), \d+, \d+, (\d+), (\d+)>", name)
    if m:
        return "igemm_nt%s<%s,%s,%s,%s>" % (m.group(1) or "", m.group(2), m.group(3), m.group(4), m.group(5))
    m = re.match(r"void rd::wgrad_tn(_split)?_kernel<(\d+), (\d+), \d+, \d+, (\d+), (\d+)>", name)
    if m:
        return "wgrad_tn%s<%s,%s,%s,%s>" % (m.group(1) or "", m.group(2), m.group(3), m.group(4), m.group(5))
    m = re.match(r"void rd::conv3_halo_split_kernel<(\d+),", name)
    if m:
        return "conv3_halo_split<%s>" % m.group(1)
    m = re.match(r"(?:void )?rd::(\w+)", name)
    if m:
        return m.group(1)[:-7] if m.group(1) == "wgrad_strip_kernel" else m.group(1)
    return name[:40]


def per_dispatch(path):
    by = collections.defaultdict(dict)
    for r in csv.DictReader(open(path)):
        d = by[r["Dispatch_Id"]]
        d[r["Counter_Name"]] = float(r["Counter_Value"])
        d["sym"] = sym(r["Kernel_Name"])
        d["dur"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return by


def mean_by_sym(by, key):
    acc = collections.defaultdict(list)
    for d in by.values():
        if key in d:
            acc[d["sym"]].append(d[key])
    return {k: sum(v) / len(v) for k, v in acc.items()}


os.makedirs(DST, exist_ok=True)
shutil.copy(os.path.join(SRC, "stats", "bench_kernel_stats.csv"), os.path.join(DST, f"{TAG}_kernel_stats.csv"))
shutil.copy(os.path.join(SRC, "bench_under_rocprof.json"), os.path.join(DST, f"{TAG}_bench_under_rocprof.json"))
stats = {}
for r in csv.DictReader(open(os.path.join(SRC, "stats", "bench_kernel_stats.csv"))):
    stats[sym(r["Name"])] = {"rocprof_name": r["Name"], "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                             "total_ms": float(r["TotalDurationNs"]) / 1e6, "pct": float(r["Percentage"])}
fetch = mean_by_sym(per_dispatch(os.path.join(SRC, "pmc_fetch", "bench_counter_collection.csv")), "FETCH_SIZE")
write = mean_by_sym(per_dispatch(os.path.join(SRC, "pmc_write", "bench_counter_collection.csv")), "WRITE_SIZE")
l2 = per_dispatch(os.path.join(SRC, "pmc_l2", "bench_counter_collection.csv"))
hit, miss = mean_by_sym(l2, "TCC_HIT_sum"), mean_by_sym(l2, "TCC_MISS_sum")
sq = per_dispatch(os.path.join(SRC, "pmc_sq", "bench_counter_collection.csv"))
gui, dur, mf = mean_by_sym(sq, "GRBM_GUI_ACTIVE"), mean_by_sym(sq, "dur"), mean_by_sym(sq, "SQ_VALU_MFMA_BUSY_CYCLES")
out = {"note": "FETCH_SIZE/WRITE_SIZE are KiB counters; fetch is doubled (gfx950 rocprofv3 tallies 128-B requests at 64 B "
               "for 16-B/lane coalesced reads, MI355X_MICROARCH.md HBM section); WRITE_SIZE is uncalibrated. "
               "GRBM_GUI_ACTIVE is summed over the 8 XCDs (clock = GUI/8/duration). MFMA pipe utilisation = "
               "SQ_VALU_MFMA_BUSY_CYCLES / (GUI/8 * 1024 SIMDs).",
       "kernels": {}}
for k, st in sorted(stats.items(), key=lambda kv: -kv[1]["total_ms"]):
    e = dict(st)
    if k in fetch and k in write:
        e["fetch_bytes_per_launch"] = 2 * fetch[k] * 1024
        e["write_bytes_per_launch"] = write[k] * 1024
        e["hbm_bytes_per_launch"] = e["fetch_bytes_per_launch"] + e["write_bytes_per_launch"]
    if k in hit:
        e["l2_hit_rate"] = hit[k] / (hit[k] + miss[k] + 1e-9)
    if k in gui and dur.get(k):
        e["clock_ghz_under_pmc"] = gui[k] / 8 / dur[k]
        if mf.get(k):
            e["mfma_pipe_util"] = mf[k] / (gui[k] / 8 * 1024)
    out["kernels"][k] = e
json.dump(out, open(os.path.join(DST, f"{TAG}_summary.json"), "w"), indent=1)
print("wrote", DST, "top kernels:")
for k, e in list(out["kernels"].items())[:8]:
    print(f"  {k:28s} calls={e['calls']:4d} avg={e['avg_us']:8.1f}us {e['pct']:5.1f}%  hbm/launch={e.get('hbm_bytes_per_launch', 0)/1e6:8.1f} MB"
          f"  L2hit={e.get('l2_hit_rate', 0):.2f} clk={e.get('clock_ghz_under_pmc', 0):.2f} mfma_util={e.get('mfma_pipe_util', 0):.2f}")
