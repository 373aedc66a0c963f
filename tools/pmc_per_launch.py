#!/usr/bin/env python3
"""Launch by launch: the counters of tools/pmc_bench.sh's passes for one kernel, in dispatch order.

A frame's extend launches differ (camera rays, then ever fewer and less coherent bounce rays), and the mean per dispatch that
tools/pmc_summary.py prints hides which of them is bound by what.  The passes are separate runs of the same command, so launch i of
one pass is launch i of the others.

  python tools/pmc_per_launch.py <dir of pmc_bench.sh> [kernel-substring] [launches to print]
"""
import csv
import glob
import os
import sys
from collections import OrderedDict


def load(root, tag, flt):
    rows = OrderedDict()
    for path in glob.glob(os.path.join(root, tag, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                if flt in r["Kernel_Name"]:
                    rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    return [rows[k] for k in sorted(rows)]


def main():
    root = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else "k_extend<false"
    most = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 30
    sq, rd, wr = load(root, "sq", flt), load(root, "rd", flt), load(root, "wr", flt)
    print(f"# {flt}: launches in dispatch order; clocks are GRBM_GUI_ACTIVE / 8 XCDs, ms at the nominal 2.4 GHz; read bytes = 128-byte + 64-byte + 32-byte")
    print("# requests of L2's memory side (Infinity-Cache hits included); VALU busy = SQ_INSTS_VALU x 4 clocks over 1024 SIMDs' clocks;")
    print("# waves/SIMD = 4 x SQ_WAVE_CYCLES over SIMD clocks (time average); wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES")
    print("launch  Mclk    ms   VALU-busy  wait  waves/SIMD  read GB  read TB/s  L2 hit")
    for i, (s, r, w) in enumerate(zip(sq, rd, wr)):
        if i >= most:
            break
        clk = s["GRBM_GUI_ACTIVE"] / 8
        sec = clk / 2.4e9
        rb = 128 * r["TCC_EA0_RDREQ_128B_sum"] + 64 * r["TCC_EA0_RDREQ_64B_sum"] + 32 * r["TCC_EA0_RDREQ_32B_sum"]
        print(f"{i:6d} {clk / 1e6:6.2f} {sec * 1e3:5.2f} {s['SQ_INSTS_VALU'] * 4 / (1024 * clk):10.2f} {s['SQ_WAIT_ANY'] / s['SQ_WAVE_CYCLES']:5.2f} "
              f"{4 * s['SQ_WAVE_CYCLES'] / (1024 * clk):11.1f} {rb / 1e9:8.2f} {rb / sec / 1e12:10.2f} {w['TCC_HIT_sum'] / (w['TCC_HIT_sum'] + w['TCC_MISS_sum']):7.2f}")


if __name__ == "__main__":
    main()
