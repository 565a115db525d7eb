#!/bin/bash
# Is the host->device staging of batch k+1 under batch k's kernels?  rocprofv3 kernel + memory-copy trace of the Trainer loop
# (bench.py's trainer_loop_measurement: resdepth_amd.Trainer.inference_one_epoch over pinned host batches):
#     bash scripts/trainer_timeline.sh [prefetch_batches]
REPO="$(pwd)"; PF="${1:-1}"; OUT="$REPO/gpurun_out/trainer_tl_$PF"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace -d "$OUT" -o t --output-format csv -- python -c "
import sys, importlib.util, torch
sys.path.insert(0, '$REPO')
spec = importlib.util.spec_from_file_location('bench_mod', '$REPO/bench.py'); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
from resdepth_amd import _lib; _lib.load()
print(b.trainer_loop_measurement(torch.device('cuda', 0), b.WORKLOADS['S'], 32, iters=24, prefetch=$PF))
" > "$OUT/out.txt" 2> "$OUT/err.txt"
tail -1 "$OUT/out.txt"
python "$REPO/scripts/copy_overlap.py" "$OUT" H2D 100
rm -f "$OUT"/*/*kernel_trace.csv "$OUT"/*/*memory_copy_trace.csv 2>/dev/null; find "$OUT" -name "*.csv" -size +1M -delete
