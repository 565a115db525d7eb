# GPU idle time inside the cfg-G inference sweep (rocprofv3 --kernel-trace of bench.py --infer): bash scripts/gaps_infer.sh
REPO="$(pwd)"; OUT="$REPO/gpurun_out/gaps_infer"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace -d "$OUT" -o g --output-format csv -- python $REPO/bench.py --infer --steps 2 --warmup 1 --no-prof > "$OUT/bench.json" 2> "$OUT/err.txt"
tail -c 400 "$OUT/bench.json"
python - "$OUT" <<'PY'
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted([(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:50]) for r in rows])
ev = ev[len(ev) // 3:]          # drop the warm-up sweep
span = ev[-1][1] - ev[0][0]
busy = 0; cs, ce = ev[0][0], ev[0][1]
gaps = []
for s, e, _ in ev[1:]:
    if s > ce:
        busy += ce - cs; gaps.append(s - ce); cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print("kernels", len(ev), "span ms %.2f busy ms %.2f idle ms %.2f (%.1f %%)" % (span / 1e6, busy / 1e6, (span - busy) / 1e6, 100 * (span - busy) / span))
gaps.sort()
print("gaps: n", len(gaps), "median us %.1f  p90 %.1f  max %.1f" % (gaps[len(gaps)//2] / 1e3, gaps[int(len(gaps)*.9)] / 1e3, gaps[-1] / 1e3))
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in ev:
    agg[n][0] += 1; agg[n][1] += e - s
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:18]:
    print("%8.2f ms %6d  %s" % (t / 1e6, c, n))
PY
python "$REPO/scripts/copy_overlap.py" "$OUT" D2H 50
