#!/bin/bash
mkdir -p gpurun_out/r06d; O=gpurun_out/r06d
timeout 3000 python -m pytest tests -q -m gpu -x > $O/gpu_tests.txt 2>&1; echo "tests rc $?"
tail -n 6 $O/gpu_tests.txt
bash scripts/profile_all.sh r06 S 2>&1 | tail -14
