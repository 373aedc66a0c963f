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
    print("# Every BASELINE.json configuration on one MI355X\n")
    print("`python bench.py --workload <name> --steps 3 --warmup 1` (20 steps for c1, the defaults for c2; `tools/run_cfgs.sh`; this table: `tools/make_configs_md.py`), one GPU, scene and queues resident in HBM.")
    print("c4 and c5 use synthetic stand-ins of the named triangle counts; their lines are the single-GPU leg of the 8-GPU configurations (every rank holds")
    print("the whole scene and traces its own tiles).  Roofs: vmem = vector-memory return rate for wave-coalesced lines (39,322 GB/s nominal; the per-lane record fetch of a traversal step is served at a quarter of it), valu = vector-ALU issue")
    print("(39.3 T lane-op/s), hbm = 8,000 GB/s; the bound named is the roof the extend kernel sits closest to.\n")
    print("The timed frames run connect of a bounce beside extend of the next one, so an extend launch shares the GPU for part of its time: the")
    print("extend columns and the roof fractions are those of the frame whose stages ran back to back (the kernel alone); `live frac` is the")
    print("bound's fraction from the extend launches of the timed frames themselves.\n")
    print("| config | workload | triangles | ms/step | Mrays/s (all rays) | extend Mrays/s | bounce-only extend Mrays/s | quad steps/ray | tri tests/ray | bound | frac "
          "| vmem frac | valu frac | hbm frac | live frac | CPU port Mrays/s (all threads) | CPU port Mrays/s (1 thread) |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for label, name in names:
        d = json.load(open(os.path.join(root, name + ".json")))
        r, c, cpu = d["roofline"], d["config"], d.get("cpu_baseline") or {}
        a = r["kernel_alone"]
        fr = a["fracs"]
        fmt = lambda v: "-" if v is None else "%.2f" % v
        print(f"| {label} | {c['workload']} | {c['triangles']} | {d['ms_per_step']:.2f} | {d['value']:.0f} | {a['extend_mrays_per_s']:.0f} | {a['bounce_extend_mrays_per_s']:.0f} | "
              f"{r['bvh_steps_per_ray']:.2f} | {r['tri_tests_per_ray']:.2f} | {a['bound']} | {a['frac']:.2f} | {fr['vmem']:.2f} | {fr['valu']:.2f} | {fr['hbm']:.3f} | "
              f"{r['frac']:.2f} ({r['bound']}) | {fmt(cpu.get('value'))} | {fmt(cpu.get('single_thread_value'))} |")
    print("\nc1 is one sample of a 512x512 frame (0.26M paths): two traversal launches that each last as long as their longest ray (about 115 us); its rate is what two such frames in flight reach.")
    print("c5's tree and triangles (1.7 GB) are beyond every cache: its bounce launches read 4.8-6.2 TB/s at L2's memory side (0.6-0.78 of the HBM peak, every request a 128-byte line) "
          "with the vector ALUs 0.7-0.9 busy (`profiles/r03_c5_extend_per_launch.txt`); the `hbm frac` column prices only the bytes a launch cannot avoid (ray records and one pass over the scene).")


if __name__ == "__main__":
    main()
