#!/usr/bin/env python3
"""One-screen summary of a bench.py JSON line:  python tools/show_bench.py <file.json>"""
import json
import sys

d = json.load(open(sys.argv[1]))
r = d["roofline"]
print(f"{d['value']:.0f} Mrays/s, {d['ms_per_step']:.3f} ms/step; stages {{{', '.join(f'{k} {v:.3f}' for k, v in d['stage_ms_per_step'].items())}}} ms")
print(f"  extend {r['extend_mrays_per_s']:.0f} (camera {r['primary_extend_mrays_per_s']:.0f}, bounce {r['bounce_extend_mrays_per_s']:.0f}) Mrays/s; "
      f"bound {r['bound']} {r['frac']:.3f} (algorithmic {r.get('frac_algorithmic', float('nan')):.3f}, lanes {r.get('lane_utilisation', {}).get('overall', float('nan')):.2f}); "
      + ", ".join(f"{k} {v['frac']:.3f}" for k, v in r["other_roofs"].items()))
c = d.get("cpu_baseline")
if c:
    print(f"  cpu baseline {c['value']:.1f} Mrays/s on {c['cores']} threads; parity_checked {c.get('parity_checked')}")
s = d.get("secondary")
if s:
    print(f"  soup: {s['ms_per_frame']:.1f} ms/frame, {s['mrays_per_s']:.0f} Mrays/s; stages {{{', '.join(f'{k} {v:.1f}' for k, v in s['stage_ms_per_frame'].items())}}} ms; "
          f"bounce extend {s['bounce_extend_mrays_per_s']:.0f}, connect {s['connect_mrays_per_s']:.0f} Mrays/s; shadow steps/ray {s['shadow_bvh_steps_per_ray']:.2f}, "
          f"tests/ray {s['shadow_tri_tests_per_ray']:.2f}; extend roofline {s['roofline']['bound']} {s['roofline']['frac']:.3f} at {s['roofline']['avg_launch_ms']:.2f} ms/launch")
