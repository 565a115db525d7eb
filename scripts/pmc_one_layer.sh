# PMC counters of one conv layer (profiling aid): bash scripts/pmc_one_layer.sh H CIN COUT fwd|dgrad|wgrad [kernel-name-substring]
set -u
REPO="$(pwd)"; OUT="$REPO/gpurun_out/pmc1"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/scripts/one_layer.py ${1:-64} ${2:-256} ${3:-128} ${4:-fwd} 3"
export KSUB="${5:-igemm}"
run() { d=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d "$OUT/$d" -o p --output-format csv -- $CMD > /dev/null 2> "$OUT/$d.err"; }
run sq SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM
run l2 TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
python - "$OUT" <<'PY'
import csv, sys, glob, collections, os
out = sys.argv[1]; ksub = os.environ["KSUB"]
agg = collections.defaultdict(float); n = 0; dur = []
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if ksub in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
f = glob.glob(out + "/sq/**/*kernel_trace.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if ksub in r["Kernel_Name"]:
        dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])); name = r["Kernel_Name"][:70]
n = len(dur); cyc = agg["GRBM_GUI_ACTIVE"] / 8 / n
print(name, "launches", n, "avg_us %.1f" % (sum(dur) / n / 1e3), "clock_GHz %.3f" % (cyc / (sum(dur) / n)))
print("mfma_busy %.3f" % (agg["SQ_VALU_MFMA_BUSY_CYCLES"] / n / 1024 / cyc),
      "lds_active %.3f" % (agg["SQ_LDS_IDX_ACTIVE"] / n / 256 / cyc), "lds_conflict %.3f" % (agg["SQ_LDS_BANK_CONFLICT"] / n / 256 / cyc),
      "wave_occupancy %.2f" % (agg["SQ_WAVE_CYCLES"] * 4 / n / 1024 / cyc),
      "wait_inst_any/wave %.3f" % (agg["SQ_WAIT_INST_ANY"] / agg["SQ_WAVE_CYCLES"]), "wait_lds/wave %.3f" % (agg["SQ_WAIT_INST_LDS"] / agg["SQ_WAVE_CYCLES"]),
      "valu_per_mfma %.2f" % ((agg["SQ_INSTS_VALU"] - agg["SQ_INSTS_MFMA"]) / agg["SQ_INSTS_MFMA"]), "lds_per_mfma %.2f" % (agg["SQ_INSTS_LDS"] / agg["SQ_INSTS_MFMA"]),
      "vmem_per_mfma %.3f" % (agg["SQ_INSTS_VMEM"] / agg["SQ_INSTS_MFMA"]), "l2_hit %.3f" % (agg["TCC_HIT_sum"] / max(1, agg["TCC_HIT_sum"] + agg["TCC_MISS_sum"])))
PY
