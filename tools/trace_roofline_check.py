#!/usr/bin/env python3
"""The bench line's launch duration against rocprofv3's own timestamps of the SAME run.

bench.py times its steps with two frames in flight and then renders a few frames one at a time, with events around their extend
launches; the roofline divides by the mean of those.  In a kernel trace of the default command these are the LAST launches of the
plain extend kernel (`k_extend<false, MODE>`; the frames in flight before them launch it four times as wide for bounce 0 and the
fused connect + extend kernel for the other bounces).  This prints both means.

  rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 bench.py --no-cpu-baseline --no-secondary > <dir>/bench.json
  python tools/trace_roofline_check.py <dir>
"""
import csv
import glob
import json
import os
import sys


def main():
    root = sys.argv[1]
    line = json.loads(open(os.path.join(root, "bench.json")).read().strip().splitlines()[-1])
    roof = line["roofline"]
    n = int(roof["launches"])
    trace = sorted(glob.glob(os.path.join(root, "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
    rows = [r for r in csv.DictReader(open(trace)) if r["Kernel_Name"].startswith("void k_extend<false,")]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    last = rows[-n:]
    mean = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last) / len(last) / 1e6
    print(f"bench line: {roof['kernel']} {roof['avg_launch_ms']:.4f} ms over {n} launches (HIP events, frames rendered one at a time after the timed steps)")
    print(f"rocprofv3 kernel trace of the same run: last {len(last)} dispatches of {last[0]['Kernel_Name'].split('(')[0]}: {mean:.4f} ms")
    print(f"ratio {mean / roof['avg_launch_ms']:.4f};  value {line['value']:.0f} {line['unit']}, {line['ms_per_step']:.3f} ms per step under the profiler")
    stats = sorted(glob.glob(os.path.join(root, "**", "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    if stats:
        print("\nper-kernel statistics of the whole run (frames in flight included: their launches share the GPU, so a launch's span is not its own):")
        for r in csv.DictReader(open(stats[-1])):
            if r["Name"].startswith("void k_") or r["Name"].startswith("k_"):
                print(f"  {r['Name'].split('(')[0]:36s} calls {int(r['Calls']):4d}  mean {float(r['AverageNs']) / 1e6:8.4f} ms  {float(r['Percentage']):6.2f} %")


if __name__ == "__main__":
    main()
