#!/bin/bash
mkdir -p gpurun_out/r06j; O=gpurun_out/r06j
for r in 1 2 3; do for K in 0 1 2 3; do
RD_DEFER_WGRAD=$K python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-secondary --no-prof 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('defer=$K round $r', d['value'], d['step_ms_median'])"; done; done | tee $O/ab.txt
