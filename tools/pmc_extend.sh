#!/bin/bash
# PMC passes over the extend microbenchmark (one rocprofv3 run per counter group).
# usage: tools/pmc_extend.sh <out-dir> [extend_bench.py args...]
set -u
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
run() {
  local tag=$1; shift
  timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$tag" -- python3 tools/extend_bench.py "${ARGS[@]}" > "$OUT/$tag.log" 2>&1 || { echo "pass $tag failed"; tail -3 "$OUT/$tag.log"; return 1; }
}
ARGS=("$@")
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU GRBM_GUI_ACTIVE GRBM_COUNT &&
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_WAVES SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum &&
run fetch FETCH_SIZE &&
python3 tools/pmc_summary.py "$OUT" k_extend
