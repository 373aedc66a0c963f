#!/bin/bash
# usage: tools/sweep.sh "<lib suffixes>" [extend_bench args]  — runs the extend microbenchmark with each library build
LIBS=$1; shift
for l in $LIBS; do
  for scene in soup1m bundled; do
    echo -n "$l $scene: "
    PT_HIP_LIB=libpt_hip$l.so timeout -k 10 200 python tools/extend_bench.py --scene $scene "$@" 2>&1 | tail -1
  done
  echo -n "$l soup1m coherent: "
  PT_HIP_LIB=libpt_hip$l.so timeout -k 10 200 python tools/extend_bench.py --scene soup1m --coherent "$@" 2>&1 | tail -1
done
