#!/bin/bash
# scripts/check_isa.sh <lib.so>: fail if the shipped code objects contain what the build is meant to exclude:
#   * packed-f32 VALU (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32): see csrc/build.sh (-fno-slp-vectorize)
#   * scratch (register spills) in any kernel
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
S=$(mktemp)
"$HERE/disasm.sh" "$1" "$S" 2>/dev/null
pk=$(grep -c -E "v_pk_(fma|mul|add)_f32" "$S" || true)
sc=$(grep -c -E "scratch_(load|store)" "$S" || true)
mf=$(grep -c "v_mfma_f32_32x32x16_bf16" "$S" || true)
mh=$(grep -c "v_mfma_f32_32x32x16_f16" "$S" || true)
rm -f "$S"
echo "check_isa: packed-f32 VALU $pk, scratch $sc, bf16 MFMA $mf, f16 MFMA $mh"
[ "$pk" = 0 ] || { echo "check_isa: packed-f32 VALU instructions in $1 (build every TU with -fno-slp-vectorize)"; exit 1; }
[ "$sc" = 0 ] || { echo "check_isa: scratch spills in $1"; exit 1; }
[ "$mf" -gt 0 ] || { echo "check_isa: no bf16 MFMA found -- disassembly failed?"; exit 1; }
[ "$mh" -gt 0 ] || { echo "check_isa: no f16 MFMA found -- the three-product bodies are missing"; exit 1; }
