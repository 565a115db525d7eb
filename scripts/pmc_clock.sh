# effective shader clock of one conv layer under different (ablation) builds: GRBM_GUI_ACTIVE / 8 XCDs / kernel duration
set -u
REPO="$(pwd)"; OUT="$REPO/gpurun_out/pmcclk"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export RD_NT_TILE=0
for m in ${MASKS:-0 55}; do
  if [ $m = 0 ]; then unset RESDEPTH_HIP_LIB; else export RESDEPTH_HIP_LIB="$REPO/ablate_libs/lib_$m.so"; fi
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d "$OUT/m$m" -o p --output-format csv -- python $REPO/scripts/one_layer.py 64 256 128 fwd 5 > /dev/null 2> "$OUT/m$m.err"
  python - "$OUT/m$m" $m <<'PY'
import csv, sys, glob, collections
d = sys.argv[1]
cc = list(csv.DictReader(open(glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0])))
kt = list(csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])))
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in kt if "igemm" in r["Kernel_Name"]]
agg = collections.defaultdict(float); n = 0
for r in cc:
    if "igemm" in r["Kernel_Name"]:
        agg[r["Counter_Name"]] += float(r["Counter_Value"])
n = len(dur)
g = agg["GRBM_GUI_ACTIVE"] / 8 / n
print("mask", sys.argv[2], "launches", n, "avg_us %.1f" % (sum(dur) / n / 1e3), "cycles/launch %.0f" % g, "clock GHz %.3f" % (g / (sum(dur) / n)),
      "mfma_busy_frac %.3f" % (agg["SQ_VALU_MFMA_BUSY_CYCLES"] / n / 1024 / g))
PY
done
