#!/bin/bash
mkdir -p gpurun_out/r06m; O=gpurun_out/r06m
timeout 900 python -m pytest tests/test_plan_gpu.py tests/test_trainer_gpu.py -q -m gpu -x > $O/t.txt 2>&1; echo "tests rc $?"; tail -n 3 $O/t.txt
for r in 1 2 3; do for X in 0 1; do RD_PLAN_SYSFENCE=$X python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-secondary --no-prof 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('sysfence=$X round $r', d['value'], d['step_ms_median'], d['host_enqueue_ms'])"; done; done | tee $O/ab.txt
bash scripts/step_timeline.sh 2>&1 | tail -3
