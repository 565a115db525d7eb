#!/bin/bash
# Build libresdepth_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="${RD_OUT:-$HERE/../libresdepth_hip.so}"      # RD_OUT / RD_OBJ / RD_EXTRA_FLAGS: variant builds for A/B runs (RESDEPTH_HIP_LIB)
OBJ="${RD_OBJ:-$HERE/obj}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
# -amdgpu-mfma-vgpr-form: accumulators stay in arch VGPRs (gfx950 has a unified file); without it hipcc shuttles all 64
# accumulator registers AGPR<->VGPR around every K-step of the weight-gradient kernel (128 extra VALU per 64 MFMA)
# -fno-slp-vectorize (every translation unit): no packed-f32 VALU (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32) anywhere.
#  (1) correctness: r02 found the last-conv weight-gradient kernel (operands from ds_read2_b32, consumed by
#      v_pk_fma_f32 ... op_sel:[0,1,0]) producing wrong LOW-half results in single 16-lane rows whenever it ran concurrently
#      with main-stream kernels of the two-stream backward (profiles/r02_notes.md "packed-FMA corruption"); no root cause,
#      so the instruction is kept out of EVERY kernel (any of them can co-run with the weight-gradient stream) and
#      scripts/check_isa.sh (called below and by tests/test_host_cpu.py) fails the build if one reappears;
#  (2) speed: packed f32 VALU beside MFMAs is an anti-lever on gfx950 (MI355X_MICROARCH.md: one v_pk_fma_f32 costs +22
#      cycles over two v_fma_f32 in an MFMA loop) -- the SLP-packed split arithmetic of the halo / NT kernels cost
#      0.3-0.6 % end to end (r03 interleaved A/B, profiles/r03_notes.md).
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -fno-slp-vectorize -Wall -Wno-unused-result $RD_EXTRA_FLAGS"
FLAGS="$BASE -mllvm -amdgpu-mfma-vgpr-form"
# RD_CLEAN=1 (what __graft_entry__.build() sets): drop every object first, so "does it build" compiles all seven translation
# units from source instead of re-linking whatever obj/ holds
[ -n "$RD_CLEAN" ] && rm -rf "$OBJ"
mkdir -p "$OBJ"
pids=()
for f in rd_runtime rd_igemm rd_convt rd_wgrad_strip rd_elementwise rd_edge_conv rd_stats; do
  if [ ! -f "$OBJ/$f.o" ] || [ "$HERE/$f.hip" -nt "$OBJ/$f.o" ] || [ "$HERE/rd_common.h" -nt "$OBJ/$f.o" ] \
     || [ "$HERE/rd_mfma_dev.h" -nt "$OBJ/$f.o" ] || [ "$HERE/rd_nt.h" -nt "$OBJ/$f.o" ] || [ "$HERE/../../include/resdepth_hip.h" -nt "$OBJ/$f.o" ]; then
    F="$FLAGS"; [ $f = rd_wgrad_strip ] && F="$BASE"     # 144 accumulator registers: AGPR-form MFMA (see the file header)
    $HIPCC $F -c "$HERE/$f.hip" -o "$OBJ/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$OBJ"/rd_runtime.o "$OBJ"/rd_igemm.o "$OBJ"/rd_convt.o "$OBJ"/rd_wgrad_strip.o "$OBJ"/rd_elementwise.o "$OBJ"/rd_edge_conv.o "$OBJ"/rd_stats.o
[ -n "$RD_SKIP_ISA_CHECK" ] || bash "$HERE/../../scripts/check_isa.sh" "$OUT"
echo "built $OUT"
