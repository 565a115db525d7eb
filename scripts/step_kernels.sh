# Kernel launches of ONE training step (between two Adam launches), grouped by name: bash scripts/step_kernels.sh
REPO="$(pwd)"; OUT="$REPO/gpurun_out/stepk"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d "$OUT" -o g --output-format csv -- python $REPO/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-secondary --no-prof > "$OUT/bench.json" 2> "$OUT/err.txt"
python - "$OUT" <<'PY'
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted([(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows])
adam = [i for i, e in enumerate(ev) if 'adam' in e[2]]
a, b = adam[-3], adam[-2]
seg = ev[a + 1:b + 1]
agg = collections.OrderedDict()
for s, e, n in seg:
    k = n.split('(')[0][-60:]
    c = agg.setdefault(k, [0, 0])
    c[0] += 1; c[1] += e - s
print("kernels in one step:", len(seg), " span ms %.3f" % ((seg[-1][1] - seg[0][0]) / 1e6))
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%4d  %8.1f us  %s" % (n, t / 1e3, k))
PY
