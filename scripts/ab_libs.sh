#!/bin/bash
# scripts/ab_libs.sh <out-dir> <rounds> <libA> <libB> ... : interleaved end-to-end A/B of library builds on ONE box
# (boxes differ by +-2 %, so only same-box alternating runs are comparable).  "-" = the in-tree library.
OUT="$1"; R="$2"; shift 2
mkdir -p "$OUT"
for r in $(seq 1 "$R"); do
  for L in "$@"; do
    tag=$(basename "$L" .so)
    if [ "$L" = "-" ]; then unset RESDEPTH_HIP_LIB; tag=base; else export RESDEPTH_HIP_LIB="$(pwd)/$L"; fi
    python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-secondary --no-prof 2>>"$OUT/err.txt" | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', 'round $r', d['value'], d['step_ms_median'])" | tee -a "$OUT/ab.txt"
  done
done
