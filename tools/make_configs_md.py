#!/usr/bin/env python3
"""Table of every benchmark configuration from a directory of bench lines.

  python tools/make_configs_md.py <dir with bundled.json c1.json c3.json c4.json c5.json> > profiles/<tag>_configs.md
"""
import json
import os
import sys


def main():
    root = sys.argv[1]
    names = [("c2 (default)", "bundled"), ("c1", "c1"), ("c3", "c3"), ("c4", "c4"), ("c5", "c5")]
    print("| config | workload | triangles | ms/step | Mrays/s (all rays) | extend Mrays/s | quad steps/ray | tri tests/ray | algorithmic GB/s (extend) "
          "| CPU port Mrays/s (all threads) | CPU port Mrays/s (1 thread) |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for label, name in names:
        d = json.load(open(os.path.join(root, name + ".json")))
        r, c, cpu = d["roofline"], d["config"], d.get("cpu_baseline") or {}
        fmt = lambda v: "-" if v is None else "%.2f" % v
        print(f"| {label} | {c['workload']} | {c['triangles']} | {d['ms_per_step']:.2f} | {d['value']:.0f} | {r['extend_mrays_per_s']:.0f} | "
              f"{r['bvh_steps_per_ray']:.2f} | {r['tri_tests_per_ray']:.2f} | {r['achieved']:.0f} | {fmt(cpu.get('value'))} | {fmt(cpu.get('single_thread_value'))} |")


if __name__ == "__main__":
    main()
