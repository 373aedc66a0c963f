#!/bin/bash
# usage: tools/run_quant_sweep.sh out.txt — extend-stage rate on incoherent bounce rays, exact against quantised node records
out=$1
for s in soup1m bundled; do
  for q in 0 1; do
    for extra in "" "--blocks 5" "--refill 24" "--refill 16"; do
      r=$(timeout -k 10 120 python tools/extend_bench.py --scene $s --rays 8000000 --iters 5 --quant $q $extra 2>&1 | grep "^extend" | cut -d, -f1-2)
      echo "$s quant=$q $extra: $r" >> $out
    done
  done
  for q in 0 1; do
    r=$(timeout -k 10 120 python tools/extend_bench.py --scene $s --rays 8000000 --iters 2 --quant $q --stats 2>&1 | grep "^extend")
    echo "$s quant=$q stats: $r" >> $out
  done
done
