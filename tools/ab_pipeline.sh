#!/bin/bash
# usage: tools/ab_pipeline.sh "spp ..." [workload]  — frames back to back with two frames in flight and with one, per sample count (same box)
W=${2:-bundled}
for spp in $1; do for flag in "" "--no-pipeline"; do
  echo -n "$W spp=$spp ${flag:-two frames in flight}: "
  timeout -k 10 300 python3 bench.py --workload $W --spp $spp $flag --steps 12 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f Mrays/s %.3f ms' % (d['value'], d['ms_per_step']))" || exit 1
done; done
