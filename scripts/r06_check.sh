#!/bin/bash
# first GPU pass of the r06 split2h arithmetic: per-op errors, whole-net tests in the mode, a quick bench of both modes
mkdir -p gpurun_out/r06a; O=gpurun_out/r06a
export RD_MFMA=split2h
timeout 600 python scripts/split_numerics.py > $O/numerics_split2h.txt 2>&1; echo "numerics rc $?"
tail -n 5 $O/numerics_split2h.txt
timeout 1500 python -m pytest tests/test_unet_gpu.py -q -m gpu -x > $O/unet_split2h.txt 2>&1; echo "unet rc $?"
tail -n 15 $O/unet_split2h.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $O/bench_split2h.json 2> $O/bench_split2h.err; echo "bench rc $?"
unset RD_MFMA
timeout 600 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $O/bench_split3.json 2> $O/bench_split3.err; echo "bench3 rc $?"
python - <<'PY'
import json
for m in ("split2h", "split3"):
    try:
        d = json.loads(open(f"gpurun_out/r06a/bench_{m}.json").read().strip().splitlines()[-1])
        print(m, d["value"], d["ms_per_step"], d.get("host_enqueue_ms"))
    except Exception as e:
        print(m, "no line", e)
PY
