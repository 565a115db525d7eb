set -u
REPO="$(pwd)"; OUT="$REPO/gpurun_out/pmc1"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/scripts/one_layer.py 64 256 128 fwd 3"
export RD_NT_TILE=0
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA -d "$OUT/sq" -o p --output-format csv -- $CMD > /dev/null 2> "$OUT/sq.err"
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM -d "$OUT/lds" -o p --output-format csv -- $CMD > /dev/null 2> "$OUT/lds.err"
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS -d "$OUT/vm" -o p --output-format csv -- $CMD > /dev/null 2> "$OUT/vm.err"
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum -d "$OUT/l2" -o p --output-format csv -- $CMD > /dev/null 2> "$OUT/l2.err"
find "$OUT" -name "*counter_collection.csv" | while read f; do echo "== $f"; python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in agg.items():
    if "igemm" in k:
        print(k, {c: v for c, v in d.items()})
PY
done
tail -3 "$OUT"/*.err
