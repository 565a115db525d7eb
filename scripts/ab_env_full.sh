#!/bin/bash
# scripts/ab_env_full.sh <rounds> <VAR=val|-> ... : as ab_env.sh, printing the secondary workloads (cfg-M, cfg-G) as well
R="$1"; shift
for r in $(seq 1 "$R"); do
  for E in "$@"; do
    if [ "$E" = "-" ]; then V=""; else V="$E"; fi
    env $V python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-prof 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['secondary']; print('$E', 'round $r', d['value'], d['step_ms_median'], 'cfg_M', s['cfg_M']['tiles_per_s'], 'cfg_G', s['cfg_G']['tiles_per_s'])"
  done
done
