for w in bundled c3 c5 c1 bundled; do for lib in libpt_hip_old.so libpt_hip.so; do
  echo -n "$w $lib: "
  PT_HIP_LIB=$lib timeout -k 10 300 python3 bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f Mrays/s %.3f ms' % (d['value'], d['ms_per_step']))" || exit 1
done; done
