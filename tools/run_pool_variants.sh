#!/bin/bash
# usage: tools/run_pool_variants.sh out.txt variant... — builds of the library with the block-level ray pool (libpt_hip_<variant>.so; "base" = libpt_hip.so):
# parity tests first (a variant that fails them is not timed), then the random-ray extend benchmark and the default frame
out=$1; shift
for v in "$@"; do
  lib=libpt_hip_$v.so; [ "$v" = base ] && lib=libpt_hip.so
  if [ "$v" != base ]; then
    PT_HIP_LIB=$lib timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 120 --timeout-method thread > gpurun_out/pool_tests_$v.log 2>&1
    rc=$?
    echo "$v parity tests: rc $rc: $(tail -1 gpurun_out/pool_tests_$v.log)" >> $out
    [ $rc -ne 0 ] && exit 1
  fi
  x=$(PT_HIP_LIB=$lib timeout -k 10 120 python tools/extend_bench.py --scene bundled --rays 8000000 --iters 5 --quant 0 2>&1 | grep "^extend" | cut -d, -f1-2) || exit 1
  echo "$v bundled random rays: $x" >> $out
  x=$(PT_HIP_LIB=$lib timeout -k 10 120 python tools/extend_bench.py --scene soup1m --rays 8000000 --iters 5 --quant 1 2>&1 | grep "^extend" | cut -d, -f1-2) || exit 1
  echo "$v soup1m random rays: $x" >> $out
  PT_HIP_LIB=$lib timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/pool_bench_$v.json 2>gpurun_out/pool_bench_$v.err || exit 1
  python tools/show_bench.py gpurun_out/pool_bench_$v.json >> $out 2>&1
done
