#!/bin/bash
# Memory-side PMC passes over the extend microbenchmark (one rocprofv3 run per counter group).
# usage: tools/pmc_mem.sh <out-dir> [extend_bench.py args...]
set -u
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
ARGS=("$@")
run() {
  local tag=$1; shift
  timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$tag" -- python3 tools/extend_bench.py "${ARGS[@]}" > "$OUT/$tag.log" 2>&1 || { echo "pass $tag failed"; tail -3 "$OUT/$tag.log"; return 1; }
}
run m1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
       TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum \
       TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum &&
run m2 TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum \
       TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE \
       SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVES SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD &&
python3 tools/pmc_summary.py "$OUT" k_extend
