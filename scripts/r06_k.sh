#!/bin/bash
mkdir -p gpurun_out/r06k; O=gpurun_out/r06k
for r in 1 2 3; do for K in 512 1024 384; do
RD_TUNE="wg_blocks=$K" python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-secondary --no-prof 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('wg_blocks=$K round $r', d['value'], d['step_ms_median'])"; done; done | tee $O/ab.txt
timeout 2400 python -m pytest tests -q -m gpu > $O/t_all.txt 2>&1; echo "all rc $?"; tail -n 6 $O/t_all.txt
