#!/bin/bash
# scripts/ab_kernels.sh <rounds> <libA> <libB> ... : interleaved A/B of library builds on one box: tiles/s and the per-step time of
# every HBM-class kernel class of the serialized pass (bench.py `kernels`).  "-" = the in-tree library.
R="$1"; shift
for r in $(seq 1 "$R"); do
  for L in "$@"; do
    tag=$(basename "$L" .so)
    if [ "$L" = "-" ]; then unset RESDEPTH_HIP_LIB; tag=tree; else export RESDEPTH_HIP_LIB="$(pwd)/$L"; fi
    python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
rows = {k['name']: k['ms_per_step'] for k in d['kernels'] if 'hbm_frac' in k}
tot = sum(rows.values())
print('$tag', 'round $r', d['value'], d['step_ms_median'], 'hbm-class total %.3f' % tot, ' '.join('%s=%.3f' % (n, v) for n, v in sorted(rows.items(), key=lambda kv: -kv[1])[:12]))"
  done
done
