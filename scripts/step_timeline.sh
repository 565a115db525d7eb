#!/bin/bash
# kernel-by-kernel timeline of ONE production step (two streams, launch plan): start / end relative to the step, stream, name
REPO="$(pwd)"; OUT="$REPO/gpurun_out/tl"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d "$OUT" -o g --output-format csv -- python $REPO/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-secondary --no-prof --diag-steps 0 > "$OUT/bench.json" 2> "$OUT/err.txt"
python - "$OUT" <<'PY'
import csv, sys, glob, re
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted([(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Stream_Id', r.get('Queue_Id', ''))) for r in rows])
adam = [i for i, e in enumerate(ev) if 'adam' in e[2]]
a, b = adam[-3], adam[-2]
seg = ev[a + 1:b + 1]
t0 = seg[0][0]
out = open(sys.argv[1] + "/timeline.txt", "w")
def short(n):
    n = re.sub(r"^void rd::", "", n); n = re.sub(r"\(.*", "", n); return n[:58]
for s, e, n, q in seg:
    out.write("%8.1f %8.1f %7.1f  q%-3s %s\n" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, short(n)))
out.close()
print("step span us", (seg[-1][1] - t0) / 1e3, "kernels", len(seg))
PY
python "$REPO/scripts/overlap_timeline.py" $(ls "$OUT"/*/*kernel_trace.csv "$OUT"/*kernel_trace.csv 2>/dev/null | head -1) --steps 4 | tail -12
