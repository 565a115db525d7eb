#!/bin/bash
# scripts/ab_env.sh <rounds> <VAR=val|-> ... : interleaved end-to-end A/B of environment settings on ONE box
R="$1"; shift
for r in $(seq 1 "$R"); do
  for E in "$@"; do
    if [ "$E" = "-" ]; then V=""; else V="$E"; fi
    env $V python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-secondary --no-prof 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$E', 'round $r', d['value'], d['step_ms_median'])"
  done
done
