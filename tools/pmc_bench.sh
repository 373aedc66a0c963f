#!/bin/bash
# Memory-side traffic of the bench's kernels: separate rocprofv3 --pmc passes (kernel-trace/stats only in their own run).
# Frames are rendered one at a time (--no-pipeline): with two frames in flight a launch shares the GPU with the other frame's kernels, and
# neither its counters nor its duration would be its own.  The roofline of the bench line divides by launch durations measured the same way.
# usage: tools/pmc_bench.sh <out-dir> [bench.py args...]
set -u
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
ARGS=("$@")
run() {
  local tag=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$tag" -- python3 bench.py --steps 2 --warmup 0 --no-pipeline --no-cpu-baseline --no-secondary "${ARGS[@]}" > "$OUT/$tag.json" 2> "$OUT/$tag.err" || { echo "pass $tag failed"; tail -3 "$OUT/$tag.err"; return 1; }
}
run rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum &&
run wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum &&
run fetch FETCH_SIZE &&
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES SQ_LEVEL_WAVES GRBM_GUI_ACTIVE &&
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --steps 3 --warmup 1 --no-pipeline --no-cpu-baseline --no-secondary "${ARGS[@]}" > "$OUT/trace.json" 2> "$OUT/trace.err"
python3 tools/pmc_summary.py "$OUT" k_extend
python3 tools/pmc_summary.py "$OUT" k_shade | grep -E "RDREQ|WRREQ|FETCH"
