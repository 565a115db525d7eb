#!/bin/bash
mkdir -p gpurun_out/r06h; O=gpurun_out/r06h
timeout 2400 python -m pytest tests -q -m gpu -k "not bench_gpus_2" > $O/t_all.txt 2>&1; echo "all rc $?"; tail -n 8 $O/t_all.txt
for L in - ab_libs/slab8.so; do if [ "$L" = "-" ]; then unset RESDEPTH_HIP_LIB; else export RESDEPTH_HIP_LIB=$(pwd)/$L; fi; for r in 1 2; do
python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
rows = {k['name']: k['ms_per_step'] for k in d['kernels']}
print('$L', d['value'], d['step_ms_median'], 'wgrad_reduce', rows.get('wgrad_reduce'))"; done; done | tee $O/slab.txt
