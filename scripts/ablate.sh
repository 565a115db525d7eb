#!/bin/bash
# Diagnosis builds of the split-bf16 NT kernel with parts removed (RD_ABLATE bitmask: 1 no split VALU, 2 no LDS stores,
# 4 no global loads, 8 one MFMA instead of six).  Results are numerically WRONG by construction; timing only.
# Builds into gpurun_out/ablate/ (run here), then on the GPU box:  bash scripts/ablate.sh run
set -e
HERE="$(cd "$(dirname "$0")/.." && pwd)"
C="$HERE/resdepth_amd/csrc"; OUT="$HERE/ablate_libs"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -mllvm -amdgpu-mfma-vgpr-form"
MASKS="${MASKS:-1 2 3 4 7 8 15}"
if [ "${1:-build}" = build ]; then
  mkdir -p "$OUT"
  for m in $MASKS; do
    /opt/rocm/bin/hipcc $FLAGS -DRD_ABLATE=$m -c "$C/rd_igemm.hip" -o "$OUT/igemm_$m.o" &
  done; wait
  for m in $MASKS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/lib_$m.so" "$C/obj/rd_runtime.o" "$OUT/igemm_$m.o" "$C/obj/rd_elementwise.o" "$C/obj/rd_stats.o"
    rm "$OUT/igemm_$m.o"
  done
  ls "$OUT"
else
  for m in 0 $MASKS; do
    if [ $m = 0 ]; then unset RESDEPTH_HIP_LIB; else export RESDEPTH_HIP_LIB="$OUT/lib_$m.so"; fi
    echo "ABLATE $m"; BL_WHICH=fwd RD_NT_TILE=${TILE:-0} timeout 100 python "$HERE/scripts/bench_layers.py" 2>&1 | grep -E "enc2|dec2|dec1"
  done
fi
