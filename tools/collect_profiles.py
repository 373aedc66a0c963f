#!/usr/bin/env python3
"""Copy a set of measured summaries from gpurun_out/ into profiles/ under a tag (run here after the GPU call).

The GPU call:   tools/pmc_bench.sh gpurun_out/final_pmc > gpurun_out/final_pmc_summary.txt
                python bench.py > gpurun_out/final_cfg/bundled.json; python bench.py --workload c1|c3|c4|c5 ... > gpurun_out/final_cfg/<w>.json
Then:           python tools/collect_profiles.py <tag>
"""
import glob
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1]
    out, prof = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
    vals = {}
    for line in open(os.path.join(out, "final_pmc_summary.txt")):
        m = re.match(r"void k_extend<false>\s+(\S+)\s+mean/dispatch\s+([\d.]+)", line)
        if m:
            vals[m.group(1)] = float(m.group(2))
    rd = 128 * vals["TCC_EA0_RDREQ_128B_sum"] + 64 * vals["TCC_EA0_RDREQ_64B_sum"] + 32 * vals["TCC_EA0_RDREQ_32B_sum"]
    wr = 64 * vals["TCC_EA0_WRREQ_64B_sum"] + 32 * (vals["TCC_EA0_WRREQ_sum"] - vals["TCC_EA0_WRREQ_64B_sum"])
    tpath = os.path.join(prof, "traffic.json")
    t = json.load(open(tpath))
    b = t["bundled"]
    b["extend_hbm_bytes_per_launch"], b["read_bytes_per_launch"], b["write_bytes_per_launch"] = rd + wr, rd, wr
    b["tcc_hit_rate"] = vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"])
    b["method"] = re.sub(r"read [\d.]+ MB in its own pass", "read %.1f MB in its own pass" % (vals["FETCH_SIZE"] * 1024 / 1e6), b["method"])
    b["valu_busy"] = vals["SQ_INSTS_VALU"] * 4 / (1024 * vals["GRBM_GUI_ACTIVE"] / 8)
    b["valu_busy_method"] = "SQ_INSTS_VALU (wave instructions, 4 cycles each on a 16-lane SIMD) over 1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs, k_extend<false>, mean per dispatch"
    json.dump(t, open(tpath, "w"), indent=1)
    shutil.copy(os.path.join(out, "final_pmc_summary.txt"), os.path.join(prof, f"{tag}_bench_bundled_pmc.txt"))
    stats = sorted(glob.glob(os.path.join(out, "final_pmc", "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
    shutil.copy(stats[-1], os.path.join(prof, f"{tag}_bench_bundled_kernel_stats.csv"))
    for w in ("c1", "c3", "c4", "c5"):
        shutil.copy(os.path.join(out, "final_cfg", w + ".json"), os.path.join(prof, f"{tag}_cfg_{w}.json"))
    d = json.load(open(os.path.join(out, "final_cfg", "bundled.json")))
    d["roofline"]["traffic"] = rd + wr          # the bench line read the previous traffic.json
    d["roofline"]["valu_busy"] = b["valu_busy"]
    open(os.path.join(prof, f"{tag}_bench_bundled.json"), "w").write(json.dumps(d) + "\n")
    table = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_configs_md.py"), os.path.join(out, "final_cfg")], capture_output=True, text=True, check=True).stdout
    cpath = os.path.join(prof, f"{tag}_configs.md")
    s = open(cpath).read()
    a, z = s.index("| config |"), s.index("\nc1 is one sample")
    open(cpath, "w").write(s[:a] + table + s[z:])
    valu_busy = vals["SQ_INSTS_VALU"] * 4 / (1024 * vals["GRBM_GUI_ACTIVE"] / 8)
    print(f"bench {d['value']:.0f} Mrays/s, {d['ms_per_step']:.2f} ms; extend launch {d['roofline']['avg_launch_ms']:.3f} ms; traffic {(rd + wr) / 1e6:.0f} MB; VALU busy {valu_busy:.2f}")


if __name__ == "__main__":
    main()
