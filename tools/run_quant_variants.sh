#!/bin/bash
# usage: tools/run_quant_variants.sh out.txt variant... — leaf-batch / stack variants of the traversal kernels, quantised records on the soup, exact on the bundled scene
out=$1; shift
for v in "$@"; do
  lib=libpt_hip_$v.so; [ "$v" = base ] && lib=libpt_hip.so
  for r in 16 24 32 40; do
    x=$(PT_HIP_LIB=$lib timeout -k 10 120 python tools/extend_bench.py --scene soup1m --rays 8000000 --iters 5 --quant 1 --refill $r 2>&1 | grep "^extend" | cut -d, -f1-2)
    echo "$v soup1m quant=1 refill=$r: $x" >> $out
  done
  x=$(PT_HIP_LIB=$lib timeout -k 10 120 python tools/extend_bench.py --scene bundled --rays 8000000 --iters 5 --quant 0 2>&1 | grep "^extend" | cut -d, -f1-2)
  echo "$v bundled quant=0: $x" >> $out
done
