REPO="$(pwd)"; OUT="$REPO/gpurun_out/gaps"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d "$OUT" -o g --output-format csv -- python $REPO/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-secondary --no-prof > "$OUT/bench.json" 2> "$OUT/err.txt"
tail -c 300 "$OUT/bench.json"
python - "$OUT" <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted([(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:40], r.get('Stream_Id','')) for r in rows])
adam = [i for i, e in enumerate(ev) if 'adam' in e[2]]
for a, b in zip(adam[-4:-1], adam[-3:]):
    seg = ev[a + 1:b + 1]
    span = seg[-1][1] - seg[0][0]
    # union of busy intervals (two streams overlap)
    busy = 0; cur_s, cur_e = seg[0][0], seg[0][1]
    for s, e, _, _ in seg[1:]:
        if s > cur_e:
            busy += cur_e - cur_s; cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print("kernels", len(seg), "span ms %.3f  any-kernel-running ms %.3f  idle ms %.3f  sum-of-durations ms %.3f" % (span / 1e6, busy / 1e6, (span - busy) / 1e6, sum(e[1] - e[0] for e in seg) / 1e6))
PY
