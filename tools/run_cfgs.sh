#!/bin/bash
# usage: tools/run_cfgs.sh <out-dir>  — the default bench line plus one line per BASELINE configuration (run on the GPU box)
set -u
OUT=$1
mkdir -p "$OUT"
timeout -k 10 400 python3 bench.py 2>/dev/null | tail -1 > "$OUT/bench.json" && python3 tools/show_bench.py "$OUT/bench.json" || exit 1
for w in c1 c3 c4 c5; do
  steps=3; [ $w = c1 ] && steps=20     # c1's frame is a fraction of a millisecond: a run long enough for consecutive frames to overlap
  timeout -k 10 400 python3 bench.py --workload $w --steps $steps --warmup 1 --no-secondary 2>/dev/null | tail -1 > "$OUT/$w.json" && echo "$w: $(python3 tools/show_bench.py "$OUT/$w.json" | head -1)" || exit 1
done
