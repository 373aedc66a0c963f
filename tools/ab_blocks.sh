#!/bin/bash
# usage: tools/ab_blocks.sh "libA.so ..." "blocks ..." [workload]  — bench lines per build of the library and per count of traversal blocks per CU (same box)
W=${3:-bundled}
for lib in $1; do for b in $2; do
  echo -n "$W $lib blocks=$b: "
  PT_HIP_LIB=$lib timeout -k 10 300 python3 bench.py --workload $W --blocks $b --steps 10 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f Mrays/s %.3f ms  stages %s' % (d['value'], d['ms_per_step'], {k: round(v,2) for k,v in d['stage_ms_per_step'].items()}))" || exit 1
done; done
