#!/usr/bin/env python3
"""Per-kernel averages of every counter collected by scripts/pmc_kernel.sh:  python scripts/pmc_kernel_summary.py <dir>"""
import csv, sys, glob, collections, json
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
durs = collections.defaultdict(list)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void rd::", "")[:90]
        per[(r["Dispatch_Id"], k)][r["Counter_Name"]] = per[(r["Dispatch_Id"], k)].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        per[(r["Dispatch_Id"], k)]["_dur"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for (_, k), d in per.items():
        for c, v in d.items():
            agg[k][c].append(v)
res = {}
for k, d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    res[k] = m
    dur = m.pop("_dur")
    print("==", k, "avg_us %.1f" % (dur / 1e3), "launches/pass", len(d["_dur"]) // max(1, len([c for c in d if c != "_dur"])) or len(d["_dur"]))
    gui = m.get("GRBM_GUI_ACTIVE")
    if gui:
        cyc = gui / 8
        print("   clock_GHz %.3f  cycles %.0f" % (cyc / dur, cyc))
        wc = m.get("SQ_WAVE_CYCLES", 0)
        if wc:
            print("   waves/SIMD %.2f  wait_inst_any/wave %.3f  wait_any/wave %.3f  active_inst_any/wave %.3f" % (
                wc * 4 / 1024 / cyc, m.get("SQ_WAIT_INST_ANY", 0) / wc, m.get("SQ_WAIT_ANY", 0) / wc, m.get("SQ_ACTIVE_INST_ANY", 0) / wc))
    for c in sorted(m):
        print("   %-44s %16.1f" % (c, m[c]))
    m["_dur_ns"] = dur
json.dump(res, open(out + "/summary.json", "w"), indent=1)
