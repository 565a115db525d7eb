#!/bin/bash
# whole GPU suite with split2h as the library default + the quick bench pair
mkdir -p gpurun_out/r06b; O=gpurun_out/r06b
timeout 3000 python -m pytest tests -q -m gpu > $O/gpu_tests.txt 2>&1; echo "tests rc $?"
tail -n 25 $O/gpu_tests.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $O/bench_split2h.json 2> $O/bench_split2h.err; echo "bench rc $?"
RD_MFMA=split3 timeout 600 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $O/bench_split3.json 2> $O/bench_split3.err; echo "bench3 rc $?"
python - <<'PY'
import json
for m in ("split2h", "split3"):
    try:
        d = json.loads(open(f"gpurun_out/r06b/bench_{m}.json").read().strip().splitlines()[-1])
        print(m, d["value"], d["ms_per_step"], d.get("host_enqueue_ms"))
    except Exception as e:
        print(m, "no line", e)
PY
