#!/bin/bash
# per-image inference slots: new tests, the eval / blend / inference tests that must not move, and the sweep rate three ways
mkdir -p gpurun_out/r06g; O=gpurun_out/r06g
timeout 900 python -m pytest tests/test_eval_per_image_gpu.py -q -m gpu -x > $O/t_img.txt 2>&1; echo "img tests rc $?"; tail -n 15 $O/t_img.txt
timeout 1500 python -m pytest tests/test_blend_gpu.py tests/test_unet_gpu.py tests/test_split2h_gpu.py tests/test_cabi_consumer_gpu.py -q -m gpu -x -k "not bench_gpus_2" > $O/t_rest.txt 2>&1; echo "rest rc $?"; tail -n 6 $O/t_rest.txt
python bench.py --infer --raster 8192 --steps 2 --warmup 1 2>/dev/null | tail -1 > $O/infer_img.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06g/infer_img.json").read())
print("cfg-G per-image:", d.get("value"), d.get("unit"), d.get("ms_per_step"))
PY
