#!/bin/bash
# Build libresdepth_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../libresdepth_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
# -amdgpu-mfma-vgpr-form: accumulators stay in arch VGPRs (gfx950 has a unified file); without it hipcc shuttles all 64
# accumulator registers AGPR<->VGPR around every K-step of the weight-gradient kernel (128 extra VALU per 64 MFMA)
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -Wall -Wno-unused-result"
FLAGS="$BASE -mllvm -amdgpu-mfma-vgpr-form"
mkdir -p "$HERE/obj"
pids=()
for f in rd_runtime rd_igemm rd_convt rd_wgrad_strip rd_elementwise rd_edge_conv rd_stats; do
  if [ ! -f "$HERE/obj/$f.o" ] || [ "$HERE/$f.hip" -nt "$HERE/obj/$f.o" ] || [ "$HERE/rd_common.h" -nt "$HERE/obj/$f.o" ] \
     || [ "$HERE/rd_mfma_dev.h" -nt "$HERE/obj/$f.o" ] || [ "$HERE/rd_nt.h" -nt "$HERE/obj/$f.o" ] || [ "$HERE/../../include/resdepth_hip.h" -nt "$HERE/obj/$f.o" ]; then
    F="$FLAGS"; [ $f = rd_wgrad_strip ] && F="$BASE"     # 144 accumulator registers: AGPR-form MFMA (see the file header)
    # rd_edge_conv: no SLP vectorisation = no v_pk_fma_f32.  With it, the last-conv weight-gradient kernel (operands
    # from ds_read2_b32, consumed by v_pk_fma_f32 ... op_sel:[0,1,0]) produced wrong LOW-half results in single 16-lane
    # rows whenever it ran concurrently with main-stream kernels of the two-stream backward (profiles/r02_notes.md,
    # "packed-FMA corruption"); the scalar-FMA build is bit-stable and equally fast (the kernels are HBM-bound)
    [ $f = rd_edge_conv ] && F="$FLAGS -fno-slp-vectorize"
    $HIPCC $F -c "$HERE/$f.hip" -o "$HERE/obj/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$HERE"/obj/rd_runtime.o "$HERE"/obj/rd_igemm.o "$HERE"/obj/rd_convt.o "$HERE"/obj/rd_wgrad_strip.o "$HERE"/obj/rd_elementwise.o "$HERE"/obj/rd_edge_conv.o "$HERE"/obj/rd_stats.o
echo "built $OUT"
