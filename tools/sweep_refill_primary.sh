for wl in c5 c3; do for rp in 48 40 32 24; do echo "== $wl refill-primary $rp"; timeout -k 10 250 python3 bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --refill-primary $rp 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'])
" || exit 1; done; done
