#!/usr/bin/env python3
"""Side-by-side table of tools/pmc_mem.sh summaries: pmc_table.py <kernel-substring> <summary.txt>..."""
import re, sys
kern = sys.argv[1]
files = sys.argv[2:]
d = {}
for f in files:
    d[f] = {}
    for l in open(f):
        m = re.search(r"(\S.*?)\s{2,}(\S+)\s+mean/dispatch\s+([\d.]+)", l)
        if m and kern in m.group(1):
            d[f][m.group(2)] = float(m.group(3))
keys = sorted(set().union(*[set(v) for v in d.values()]))
print("%-42s" % "counter" + "".join("%18s" % f.split("/")[-1][:17] for f in files))
for k in keys:
    print("%-42s" % k + "".join("%18.0f" % d[f].get(k, float("nan")) for f in files))
for f in files:
    v = d[f]
    g = v["GRBM_GUI_ACTIVE"] / 8
    print(f.split("/")[-1], "cycles/XCD %.2fM" % (g / 1e6), "| TCC hit %.3f" % (v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])),
          "| TCC read latency %.0f" % (v["TCP_TCC_READ_REQ_LATENCY_sum"] / v["TCP_TCC_READ_REQ_sum"]),
          "| TCP hit %.3f" % (1 - v["TCP_TCC_READ_REQ_sum"] / v["TCP_TOTAL_CACHE_ACCESSES_sum"]),
          "| TA busy %.2f" % (v["TA_TA_BUSY_sum"] / 256 / g), "| TD busy %.2f" % (v["TD_TD_BUSY_sum"] / 256 / g),
          "| TCP accesses/clk/CU %.2f" % (v["TCP_TOTAL_CACHE_ACCESSES_sum"] / 256 / g),
          "| VALU issue %.2f" % (v["SQ_INSTS_VALU"] * 4 / (1024 * g)), "| SALU/VALU %.2f" % (v["SQ_INSTS_SALU"] / v["SQ_INSTS_VALU"]),
          "| lanes/VALU %.1f" % (v["SQ_THREAD_CYCLES_VALU"] / v["SQ_ACTIVE_INST_VALU"] if v.get("SQ_ACTIVE_INST_VALU") else 0),
          "| waves %.0f" % v["SQ_WAVES"], "| waves/SIMD, time average (SQ_WAVE_CYCLES ticks every 4th clock) %.1f" % (4 * v["SQ_WAVE_CYCLES"] / (1024 * g)))
