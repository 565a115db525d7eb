#!/usr/bin/env python3
"""Condense rocprofv3 output (scripts/profile.sh -> gpurun_out/prof/) into profiles/r<NN>_*:
  r01_kernel_stats.csv      kernel-trace --stats summary of `bench.py` (per kernel symbol: calls, avg, total, %)
  r01_summary.json          per kernel symbol: avg duration, HBM bytes per launch from the separate --pmc passes
                            (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16-byte coalesced reads on
                            gfx950, WRITE_SIZE as reported; both in KiB units of the counter), L2 hit rate,
                            effective clock and MFMA pipe utilisation."""
import collections, csv, json, os, re, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r01"
SRC = os.path.join(ROOT, "gpurun_out", sys.argv[2] if len(sys.argv) > 2 else "prof")      # PROF_DIR of scripts/profile.sh
DST = os.environ.get("PROF_DST") or os.path.join(ROOT, "profiles")     # PROF_DST: summarise on the GPU box into gpurun_out/


def sym(name):
    """rocprof kernel name -> key of the summary: the symbol WITH its template arguments (two instantiations of one kernel
    -- bn_act_bwd_kernel<true, true> / <false, true> -- are different rows; r02 keyed by the bare name and one overwrote the
    other), namespace / `void` / argument list stripped, the long NT / TN argument lists abbreviated as rd_prof prints them."""
    name = name.strip()
    m = re.match(r"void rd::igemm_nt(_split)?_kernel<(\d+), (\d+), \d+, \d+, (\d+), (\d+)>", name)
    if m:
        return "igemm_nt%s<%s,%s,%s,%s>" % (m.group(1) or "", m.group(2), m.group(3), m.group(4), m.group(5))
    m = re.match(r"void rd::wgrad_tn(_split)?_kernel<(\d+), (\d+), \d+, \d+, (\d+), (\d+)>", name)
    if m:
        return "wgrad_tn%s<%s,%s,%s,%s>" % (m.group(1) or "", m.group(2), m.group(3), m.group(4), m.group(5))
    m = re.match(r"void rd::conv3_halo_split_kernel<(\d+), \d+, \d+, (\d+), \d+, (\d+)>", name)
    if m:      # the two-images-per-patch instantiation (last argument 1) is its own row, as rd_prof prints it
        return "conv3_halo_split<%s%s>" % (m.group(1), ",w8" if m.group(3) == "1" else "")
    m = re.match(r"void rd::conv3_halo_split_kernel<(\d+), \d+, \d+, (\d+)", name)
    if m:
        return "conv3_halo_split<%s>" % m.group(1)
    m = re.match(r"(?:void )?rd::(\w+?)(?:_kernel)?(<[^(]*>)?\(", name) or re.match(r"(?:void )?rd::(\w+?)(?:_kernel)?(<.*>)?$", name)
    if m:
        return m.group(1) + (m.group(2) or "").replace(" ", "")
    return name[:60]


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
n_mfma = mean_by_sym(sq, "SQ_INSTS_MFMA")
lds_path = os.path.join(SRC, "pmc_lds", "bench_counter_collection.csv")
lds = per_dispatch(lds_path) if os.path.exists(lds_path) else {}
n_valu, n_lds, n_vmem = mean_by_sym(lds, "SQ_INSTS_VALU"), mean_by_sym(lds, "SQ_INSTS_LDS"), mean_by_sym(lds, "SQ_INSTS_VMEM")
bank, lds_act = mean_by_sym(lds, "SQ_LDS_BANK_CONFLICT"), mean_by_sym(lds, "SQ_LDS_IDX_ACTIVE")
out = {"note": "FETCH_SIZE/WRITE_SIZE are KiB counters; fetch is doubled (gfx950 rocprofv3 tallies 128-B requests at 64 B "
               "for 16-B/lane coalesced reads, MI355X_MICROARCH.md HBM section); WRITE_SIZE is uncalibrated. "
               "GRBM_GUI_ACTIVE is summed over the 8 XCDs (clock = GUI/8/duration). MFMA pipe utilisation = "
               "SQ_VALU_MFMA_BUSY_CYCLES / (GUI/8 * 1024 SIMDs).",
       "kernels": {}}
# the library and the arithmetic the passes ran with (bench.py quotes `traffic` / `pmc` from a summary only when both match
# the run that reads it): from the bench line captured under the kernel trace, else from the library as it loads here
try:
    line = json.loads(open(os.path.join(SRC, "bench_under_rocprof.json")).read().strip().splitlines()[-1])
    out["arithmetic_mode"] = line.get("arithmetic_mode")
except Exception:       # noqa: BLE001
    out["arithmetic_mode"] = None
try:
    sys.path.insert(0, ROOT)
    from resdepth_amd import _lib
    out["rd_version"] = _lib.load().rd_version()
    if out["arithmetic_mode"] is None:
        out["arithmetic_mode"] = _lib.mfma_mode()
except Exception:       # noqa: BLE001
    out["rd_version"] = None
for k, st in sorted(stats.items(), key=lambda kv: -kv[1]["total_ms"]):
    e = dict(st)
    if k in fetch and k in write:
        e["fetch_bytes_per_launch"] = 2 * fetch[k] * 1024
        e["write_bytes_per_launch"] = write[k] * 1024
        e["hbm_bytes_per_launch"] = e["fetch_bytes_per_launch"] + e["write_bytes_per_launch"]
    if k in hit:
        e["l2_hit_rate"] = hit[k] / (hit[k] + miss[k] + 1e-9)
    if k in gui and dur.get(k):
        # GRBM_GUI_ACTIVE / 8 XCDs / duration is an effective clock only when the kernel fills the chip for most of its
        # duration: below ~100 us the ramp-up / drain and the counter's own granularity dominate (r02 showed 2.6-13 "GHz")
        if dur[k] >= 100e3:
            e["clock_ghz_under_pmc"] = gui[k] / 8 / dur[k]
        if mf.get(k):
            e["mfma_pipe_util"] = mf[k] / (gui[k] / 8 * 1024)
    if n_mfma.get(k) and k in n_valu:      # instruction mix per MFMA (both counters are per-wave instruction counts)
        e["valu_per_mfma"] = n_valu[k] / n_mfma[k]
        e["lds_inst_per_mfma"] = n_lds.get(k, 0.0) / n_mfma[k]
        e["vmem_inst_per_mfma"] = n_vmem.get(k, 0.0) / n_mfma[k]
    if lds_act.get(k):
        e["lds_bank_conflict_frac"] = bank.get(k, 0.0) / lds_act[k]
    out["kernels"][k] = e
json.dump(out, open(os.path.join(DST, f"{TAG}_summary.json"), "w"), indent=1)
print("wrote", DST, "top kernels:")
for k, e in list(out["kernels"].items())[:8]:
    print(f"  {k:28s} calls={e['calls']:4d} avg={e['avg_us']:8.1f}us {e['pct']:5.1f}%  hbm/launch={e.get('hbm_bytes_per_launch', 0)/1e6:8.1f} MB"
          f"  L2hit={e.get('l2_hit_rate', 0):.2f} clk={e.get('clock_ghz_under_pmc', 0):.2f} mfma_util={e.get('mfma_pipe_util', 0):.2f}")
