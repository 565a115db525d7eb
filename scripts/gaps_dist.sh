# GPU busy/idle analysis of the DP (RCCL, world 1) bench run
REPO="$(pwd)"; OUT="$REPO/gpurun_out/gapsd"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d "$OUT" -o g --output-format csv -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 $REPO/bench.py --gpus 1 --steps 8 --warmup 3 --force-dist --no-cpu-baseline --no-secondary --no-prof > "$OUT/bench.json" 2> "$OUT/err.txt"
tail -c 200 "$OUT/bench.json"
python - "$OUT" <<'PY'
import csv, sys, glob, collections
fs = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in fs:
    rows += list(csv.DictReader(open(f)))
ev = sorted([(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:60]) for r in rows])
adam = [i for i, e in enumerate(ev) if 'adam' in e[2]]
a, b = adam[-3], adam[-2]
seg = ev[a + 1:b + 1]
span = seg[-1][1] - seg[0][0]
busy = 0; cs, ce = seg[0][0], seg[0][1]; gaps = []
for s, e, n in seg[1:]:
    if s > ce:
        gaps.append((s - ce, n)); busy += ce - cs; cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print("kernels", len(seg), "span ms %.3f busy %.3f idle %.3f" % (span / 1e6, busy / 1e6, (span - busy) / 1e6))
for g, n in sorted(gaps, reverse=True)[:8]:
    print("  gap %.1f us before %s" % (g / 1e3, n))
dur = collections.defaultdict(float)
for s, e, n in seg:
    dur[n[:40]] += (e - s) / 1e6
for n, d in sorted(dur.items(), key=lambda kv: -kv[1])[:8]:
    print("  %-42s %.3f ms" % (n, d))
PY
