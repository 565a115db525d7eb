#!/bin/bash
# rocprofv3 evidence for profiles/: (1) kernel-trace + stats of the bench command, (2..) PMC passes, each alone.
# Run on the GPU box from the repo root:  bash scripts/profile.sh
#   PROF_ARGS="--workload M"            cfg-M (2-ch 512x512 depth-6, batch 32)     PROF_DIR=prof_cfgM
#   PROF_ARGS="--infer --raster 8192"   cfg-G (tiled inference sweep, 3969 tiles)  PROF_DIR=prof_cfgG
# then on the build host:  python scripts/summarize_prof.py r04_cfgM prof_cfgM
set -u
REPO="$(pwd)"
OUT="$REPO/gpurun_out/${PROF_DIR:-prof}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="${PROF_ARGS:-}"
STEPS="${PROF_STEPS:-5}"
BENCH="python $REPO/bench.py --steps $STEPS --warmup 2 --no-cpu-baseline --no-secondary --serial-backward $ARGS"
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o bench --output-format csv -- $BENCH > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err"
PMCBENCH="python $REPO/bench.py --steps ${PMC_STEPS:-2} --warmup 1 --no-cpu-baseline --no-secondary --no-prof --serial-backward $ARGS"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA -d "$OUT/pmc_sq" -o bench --output-format csv -- $PMCBENCH > /dev/null 2> "$OUT/pmc_sq.err"
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM -d "$OUT/pmc_lds" -o bench --output-format csv -- $PMCBENCH > /dev/null 2> "$OUT/pmc_lds.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o bench --output-format csv -- $PMCBENCH > /dev/null 2> "$OUT/pmc_fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_write" -o bench --output-format csv -- $PMCBENCH > /dev/null 2> "$OUT/pmc_write.err"
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d "$OUT/pmc_l2" -o bench --output-format csv -- $PMCBENCH > /dev/null 2> "$OUT/pmc_l2.err"
find "$OUT" -name "*.csv" | xargs ls -la
