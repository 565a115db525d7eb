#!/bin/bash
# scripts/ab_tune.sh <out-dir> <rounds> <RD_TUNE-A> <RD_TUNE-B> ... : interleaved end-to-end A/B of tuning-knob settings on ONE box
# ("-" = defaults).  Same idea as ab_libs.sh: boxes differ by +-2 %, only alternating same-box runs are comparable.
OUT="$1"; R="$2"; shift 2
mkdir -p "$OUT"
for r in $(seq 1 "$R"); do
  for T in "$@"; do
    if [ "$T" = "-" ]; then unset RD_TUNE; else export RD_TUNE="$T"; fi
    python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-secondary --no-prof 2>>"$OUT/err.txt" | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$T', 'round $r', d['value'], d['step_ms_median'])" | tee -a "$OUT/ab_tune.txt"
  done
done
