#!/bin/bash
# usage: run_bench_variants.sh out.txt variant...   (variant "base" = libpt_hip.so; "name:args" passes extra bench.py arguments)
out=$1; shift
for v in "$@"; do
  name=${v%%:*}; extra=""; [ "$v" != "$name" ] && extra=${v#*:}
  lib=libpt_hip_$name.so; [ -f pathtracer_amd/$lib ] || lib=libpt_hip.so
  r=$(PT_HIP_LIB=$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --steps 5 $extra 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print(round(d['value']), round(d['ms_per_step'],3), 'ext', round(r['extend_mrays_per_s']), 'bounce', round(r['bounce_extend_mrays_per_s']), 'prim', round(r['primary_extend_mrays_per_s']), d['stage_ms_per_step'])")
  echo "$v $r" >> $out
done
