#!/bin/bash
# usage: run_wl_variants.sh out.txt workload "args" ...   (bench.py --workload <w> with extra arguments, one line per variant)
out=$1; w=$2; shift; shift
for v in "$@"; do
  r=$(timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --no-secondary --steps 3 $v 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print(round(d['value']), round(d['ms_per_step'],2), 'ext', round(r['extend_mrays_per_s']), 'bounce', round(r['bounce_extend_mrays_per_s']), 'prim', round(r['primary_extend_mrays_per_s']), {k: round(x, 2) for k, x in d['stage_ms_per_step'].items()})")
  echo "$w [$v] $r" >> $out
done
