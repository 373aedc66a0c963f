#!/bin/bash
# usage: tools/ab_libs_args.sh "libA.so libB.so ..." "bench.py arguments" [steps]  — the same bench command with each build of the library (same box)
for lib in $1; do
  echo -n "$2 $lib: "
  PT_HIP_LIB=$lib timeout -k 10 300 python3 bench.py $2 --steps ${3:-4} --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f Mrays/s %.3f ms' % (d['value'], d['ms_per_step']))" || exit 1
done
