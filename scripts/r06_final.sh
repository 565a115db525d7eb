#!/bin/bash
# final measurements of the round: graph / plan / trainer tests (rd_copy_segments), the default bench line with its secondaries and
# CPU baseline, the cfg-G line
mkdir -p gpurun_out/r06final; O=gpurun_out/r06final
timeout 1200 python -m pytest tests/test_graph_gpu.py tests/test_plan_gpu.py tests/test_trainer_gpu.py tests/test_engine_state_gpu.py -q -m gpu -x > $O/t.txt 2>&1; echo "tests rc $?"; tail -n 4 $O/t.txt
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
timeout 600 python bench.py --infer --raster 8192 --steps 2 --warmup 1 > $O/cfgG_bench.json 2> $O/cfgG.err; echo "cfgG rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06final/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "host", d.get("host_enqueue_ms"), "roof", d["roofline"]["frac"], d["roofline"].get("traffic"), "cpu", d["cpu_baseline"]["value"])
s = d.get("secondary", {})
for k, v in s.items():
    if isinstance(v, dict):
        print(" ", k, {kk: vv for kk, vv in v.items() if isinstance(vv, (int, float))})
g = json.loads(open("gpurun_out/r06final/cfgG_bench.json").read().strip().splitlines()[-1])
print("cfgG", g["value"], g["ms_per_step"], g["roofline"]["frac"], g["roofline"].get("traffic"))
PY
