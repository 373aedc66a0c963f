#!/usr/bin/env python3
"""Register, LDS and scratch use of the traversal and shading kernels, read from the code object inside libpt_hip.so
(llvm-readelf --notes on the gfx950 bundle): the figures the occupancy statements in the design notes rest on.

  python tools/kernel_resources.py [kernel-substring ...]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    want = sys.argv[1:] or ["k_extend", "k_connect", "k_shade", "k_generate", "k_trace"]
    lib = os.path.join(ROOT, "pathtracer_amd", os.environ.get("PT_HIP_LIB", "libpt_hip.so"))   # PT_HIP_LIB: another build of the library
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "co.o")
        subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", lib], check=True)
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    for block in notes.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block)
        if not name or not any(w in name.group(1) for w in want):
            continue
        get = lambda key: (re.search(rf"\.{key}:\s+(\d+)", block) or [None, "?"])[1]
        demangled = subprocess.run(["c++filt", name.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
        print(f"{demangled:28s} vgpr {get('vgpr_count'):>3s}  sgpr {get('sgpr_count'):>3s}  spilled vgpr {get('vgpr_spill_count'):>2s}  "
              f"lds {get('group_segment_fixed_size'):>6s} B  scratch {get('private_segment_fixed_size'):>4s} B  max block {get('max_flat_workgroup_size')}")


if __name__ == "__main__":
    main()
