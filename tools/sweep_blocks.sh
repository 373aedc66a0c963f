#!/bin/bash
# Resident traversal blocks per CU against every benchmark configuration: one bench line per (workload, blocks).
# usage: tools/sweep_blocks.sh [out-file]
out=${1:-gpurun_out/sweep_blocks.txt}
mkdir -p "$(dirname "$out")"
: > "$out"
for wl in bundled c3 c4 c5; do
  for b in 4 5 6 7; do
    timeout -k 10 240 python3 bench.py --workload $wl --blocks $b --steps 2 --warmup 1 --no-cpu-baseline > /tmp/line.json 2>/tmp/err.txt || { echo "$wl $b failed" >> "$out"; tail -5 /tmp/err.txt >> "$out"; exit 1; }
    python3 - "$wl" "$b" >> "$out" <<'PY'
import json, sys
d = json.loads(open("/tmp/line.json").read().strip().splitlines()[-1])
print(sys.argv[1], "blocks", sys.argv[2], "Mrays/s %.0f" % d["value"], "ms/step %.2f" % d["ms_per_step"], {k: round(v, 2) for k, v in d.get("stage_ms_per_step", {}).items()})
PY
  done
done
