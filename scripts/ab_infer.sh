#!/bin/bash
# scripts/ab_infer.sh <out-dir> <rounds> <libA> <libB> ... : interleaved A/B of library builds on the cfg-G sweep (one box).  "-" = in-tree.
OUT="$1"; R="$2"; shift 2
mkdir -p "$OUT"
for r in $(seq 1 "$R"); do
  for L in "$@"; do
    tag=$(basename "$L" .so)
    if [ "$L" = "-" ]; then unset RESDEPTH_HIP_LIB; tag=base; else export RESDEPTH_HIP_LIB="$(pwd)/$L"; fi
    python bench.py --infer --steps 3 --warmup 1 --no-prof 2>>"$OUT/err.txt" | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', 'round $r', d['value'], d['ms_per_step'], d['raster_checksum'])" | tee -a "$OUT/ab.txt"
  done
done
