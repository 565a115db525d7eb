#!/bin/bash
# rocprofv3 evidence for profiles/: (1) kernel-trace + stats of the bench command, (2..) PMC passes, each alone.
# Run on the GPU box from the repo root:  bash scripts/profile.sh
set -u
REPO="$(pwd)"
OUT="$REPO/gpurun_out/prof"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --serial-backward"
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o bench --output-format csv -- $BENCH > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err"
PMCBENCH="python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-prof --serial-backward"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA -d "$OUT/pmc_sq" -o bench --output-format csv -- $PMCBENCH > /dev/null 2> "$OUT/pmc_sq.err"
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM -d "$OUT/pmc_lds" -o bench --output-format csv -- $PMCBENCH > /dev/null 2> "$OUT/pmc_lds.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o bench --output-format csv -- $PMCBENCH > /dev/null 2> "$OUT/pmc_fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_write" -o bench --output-format csv -- $PMCBENCH > /dev/null 2> "$OUT/pmc_write.err"
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d "$OUT/pmc_l2" -o bench --output-format csv -- $PMCBENCH > /dev/null 2> "$OUT/pmc_l2.err"
find "$OUT" -name "*.csv" | xargs ls -la
