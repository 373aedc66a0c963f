#!/bin/bash
# usage: run_variants.sh out.txt variant...
out=$1; shift
for v in "$@"; do
  lib=libpt_hip_$v.so; [ "$v" = base ] && lib=libpt_hip.so
  for s in bundled soup1m; do
    r=$(PT_HIP_LIB=$lib timeout -k 10 120 python tools/extend_bench.py --scene $s --rays 8000000 --iters 5 2>&1 | grep "^extend" | cut -d, -f1-2)
    echo "$v $s $r" >> $out
  done
done
