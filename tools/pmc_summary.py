#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: mean counter value per dispatch for kernels matching a filter.

  python tools/pmc_summary.py <dir> [kernel-substring]
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    acc = defaultdict(lambda: [0.0, 0])
    dur = defaultdict(lambda: [0.0, 0])
    meta = {}
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                name = row["Kernel_Name"]
                if flt not in name:
                    continue
                short = name.split("(")[0][-40:]
                k = (short, row["Counter_Name"])
                acc[k][0] += float(row["Counter_Value"])
                acc[k][1] += 1
                meta[short] = (row.get("VGPR_Count"), row.get("Accum_VGPR_Count"), row.get("SGPR_Count"), row.get("LDS_Block_Size"), row.get("Scratch_Size"))
                if "Start_Timestamp" in row and row["Start_Timestamp"]:
                    dur[short][0] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                    dur[short][1] += 1
    # register counts are not printed: rocprofv3's VGPR_Count column is not the kernel's register count on gfx950
    # (tools/kernel_resources.py reads the real figures from the code object)
    for short, m in sorted(meta.items()):
        print(f"{short}: lds={m[3]} scratch={m[4]}")
    for (short, counter), (total, n) in sorted(acc.items()):
        print(f"{short:42s} {counter:36s} mean/dispatch {total / n:18.1f}  dispatches {n}")


if __name__ == "__main__":
    main()
