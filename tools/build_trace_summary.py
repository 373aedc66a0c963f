#!/usr/bin/env python3
"""Per-kernel and per-level times of one GPU BVH build, from a rocprofv3 --kernel-trace database.

  rocprofv3 --kernel-trace -d DIR -o bt -- python3 tools/build_bench.py --scene c5
  python tools/build_trace_summary.py DIR/bt_results.db [build_number]

A build starts at k_prims and ends at k_gather; a level starts at k_split_begin.
"""
import collections
import sqlite3
import sys


def short(name):
    n = name.split("(")[0]
    return n.split("::")[-1][:32] if "ptbuild::" in n else ("rocprim " + n[-24:] if "rocprim" in n.lower() or "ROCPRIM" in n else n[-32:])


def main():
    db = sqlite3.connect(sys.argv[1])
    which = int(sys.argv[2]) if len(sys.argv) > 2 else -1
    rows = list(db.execute("select name,start,end from kernels order by start"))
    starts = [i for i, r in enumerate(rows) if "k_prims" in r[0]]
    ends = [i for i, r in enumerate(rows) if "ptbuild::k_gather" in r[0]]
    builds = [(a, min(e for e in ends if e > a)) for a in starts if any(e > a for e in ends)]
    a, b = builds[which]
    seg = rows[a:b + 1]
    tot = collections.OrderedDict()
    for n, s, e in seg:
        k = short(n)
        tot.setdefault(k, [0, 0])
        tot[k][0] += e - s
        tot[k][1] += 1
    print(f"build {which}: {len(seg)} launches, kernels {sum(v[0] for v in tot.values()) / 1e6:.2f} ms, first start to last end {(seg[-1][2] - seg[0][1]) / 1e6:.2f} ms")
    for k, v in sorted(tot.items(), key=lambda x: -x[1][0]):
        print(f"  {v[0] / 1e6:8.3f} ms {v[1]:4d}  {k}")
    levels, cur = [], None
    for n, s, e in seg:
        if "k_split_begin" in n:
            cur = {"start": s, "k": collections.Counter(), "busy": 0}
            levels.append(cur)
        if "k_small_subtrees" in n:
            cur = None
        if cur is not None:
            cur["k"][short(n)] += e - s
            cur["busy"] += e - s
            cur["end"] = e
    for i, l in enumerate(levels):
        top = ", ".join(f"{k} {v / 1e6:.2f}" for k, v in l["k"].most_common(4))
        print(f"  level {i:2d}: span {(l['end'] - l['start']) / 1e6:6.2f} ms, kernels {l['busy'] / 1e6:6.2f} ms: {top}")


if __name__ == "__main__":
    main()
