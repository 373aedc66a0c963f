#!/bin/bash
# usage: tools/ab_env.sh VAR "v1 v2 ..." "workload ..." [steps] [warmup]  — bench lines of every workload with VAR set to each value (run on the GPU box)
VAR=$1; VALUES=$2; WORKLOADS=$3
for w in $WORKLOADS; do for v in $VALUES; do
  echo -n "$w $VAR=$v: "
  env $VAR=$v timeout -k 10 300 python3 bench.py --workload $w --steps ${4:-3} --warmup ${5:-1} --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f Mrays/s %.3f ms' % (d['value'], d['ms_per_step']))" || exit 1
done; done
