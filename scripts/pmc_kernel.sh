#!/bin/bash
# Every counter we can get for ONE kernel family of the training step (what bounds it?):
#     bash scripts/pmc_kernel.sh <kernel-name-regex> <tag> [bench args]
# Runs `bench.py --serial-backward` under rocprofv3 once per counter group (PMC alone with --kernel-trace, as gpurun requires),
# restricted to kernels matching the regex, and prints per-kernel averages of every counter + derived ratios.  Counter names
# the box does not offer (rocprofv3 -L) are dropped from the groups instead of failing the pass.
set -u
REGEX="${1:-conv_first_wgrad}"; TAG="${2:-k}"; shift; shift
REPO="$(pwd)"; OUT="$REPO/gpurun_out/pmck_$TAG"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
[ -f "$REPO/gpurun_out/counters_avail.txt" ] || rocprofv3 -L > "$REPO/gpurun_out/counters_avail.txt" 2>&1
AVAIL="$REPO/gpurun_out/counters_avail.txt"
CMD="python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-prof --serial-backward $*"
have() { grep -qw "$1" "$AVAIL"; }
run() {
  d=$1; shift; L=""
  case " ${PMC_GROUPS:-g1 g2 g3 g4 g5 g6 g7 g8 g9} " in *" $d "*) ;; *) return ;; esac      # PMC_GROUPS="g3 g4": only those passes
  for c in "$@"; do if have "$c"; then L="$L $c"; else echo "counter $c: not offered by this box" >> "$OUT/dropped.txt"; fi; done
  [ -z "$L" ] && return
  rocprofv3 --kernel-trace --kernel-include-regex "$REGEX" --pmc $L -d "$OUT/$d" -o p --output-format csv -- $CMD > /dev/null 2> "$OUT/$d.err"
}
run g1 GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_BUSY_CU_CYCLES
run g2 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM
run g3 SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL
run g4 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA
run g5 TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
run g6 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_TAG_STALL_sum TCC_BUSY_sum
run g7 TA_BUSY_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
run g8 FETCH_SIZE
run g9 WRITE_SIZE
python "$REPO/scripts/pmc_kernel_summary.py" "$OUT"
rm -rf "$OUT"/g*/   # raw CSVs are large; the summary and the .err files stay
