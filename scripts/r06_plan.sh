#!/bin/bash
mkdir -p gpurun_out/r06c; O=gpurun_out/r06c
timeout 1500 python -m pytest tests/test_plan_gpu.py tests/test_graph_gpu.py -q -m gpu -x > $O/plan_tests.txt 2>&1; echo "tests rc $?"
tail -n 30 $O/plan_tests.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $O/bench_plan.json 2> $O/bench_plan.err; echo "bench rc $?"
tail -3 $O/bench_plan.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06c/bench_plan.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "host", d.get("host_enqueue_ms"), "host eager", d.get("host_enqueue_ms_eager"))
print("plan", d.get("launch_plan"))
print("eager", d.get("eager_iteration"))
PY
