#!/bin/bash
# One gpurun call: rocprofv3 evidence for the three measured workloads (cfg-S headline, cfg-M, cfg-G), summarised ON the box
# (the raw counter CSVs of the cfg-G sweep exceed what gpurun merges back) into gpurun_out/profiles/ -- copy from there into
# profiles/.   bash scripts/profile_all.sh r04 [S] [M] [G]
TAG="${1:-r04}"; shift
WHAT="${*:-S M G}"
export PROF_DST="$(pwd)/gpurun_out/profiles"
mkdir -p "$PROF_DST"
for w in $WHAT; do
  case $w in
    S) PROF_DIR=prof PROF_ARGS="" bash scripts/profile.sh > /dev/null; python scripts/summarize_prof.py "$TAG" prof | tail -12 ;;
    M) PROF_DIR=prof_cfgM PROF_ARGS="--workload M" PROF_STEPS=3 bash scripts/profile.sh > /dev/null; python scripts/summarize_prof.py "${TAG}_cfgM" prof_cfgM | tail -12 ;;
    G) PROF_DIR=prof_cfgG PROF_ARGS="--infer --raster 8192" PROF_STEPS=2 PMC_STEPS=1 bash scripts/profile.sh > /dev/null; python scripts/summarize_prof.py "${TAG}_cfgG" prof_cfgG | tail -12
       rm -rf gpurun_out/prof_cfgG/pmc_* gpurun_out/prof_cfgG/stats/*trace* ;;
  esac
done
ls -la "$PROF_DST"
