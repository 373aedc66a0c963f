#!/bin/bash
# usage: tools/ab_libs.sh "libA.so libB.so ..." "workload ..." [steps]  — bench lines of every workload with each build of the library (same box)
for w in $2; do for lib in $1; do
  echo -n "$w $lib: "
  PT_HIP_LIB=$lib timeout -k 10 300 python3 bench.py --workload $w --steps ${3:-4} --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f Mrays/s %.3f ms  shade %.3f' % (d['value'], d['ms_per_step'], d['stage_ms_per_step']['shade']))" || exit 1
done; done
