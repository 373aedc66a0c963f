#!/usr/bin/env python3
"""Vector-ALU instructions of one traversal iteration, counted in the gfx950 assembly of the kernels themselves.

The traversal kernels are bound by vector-ALU issue, so their roofline is stated in lane-operations: one interior step
costs I_int VALU instructions, one triangle test I_leaf.  Both are properties of the source; this tool derives them
instead of trusting a hand count.  pt_api.hip is compiled to assembly with -DPT_ISA_MARKS, which brackets one interior
iteration and one leaf phase with comment markers; the body of a bracket is every basic block that lies on a path from
its begin marker to its end marker (forward reachability from `begin` that stops at `end`, intersected with backward
reachability from `end` that stops at `begin`), so loop rotation and block placement by the compiler do not matter.
Inner loops of a body (the stack pop, the dealing of triangles to lanes) are counted once; the part of the interior step that runs
only when a ray's stack leaves LDS (bracketed by cold markers) is left out.  Cross-check against the hardware: the wave iterations
the instrumented kernel counts times these costs should come close to SQ_INSTS_VALU of the same launch (tools/pmc_bench.sh).

  python tools/isa_cost.py            # prints the table and writes pathtracer_amd/isa_cost.json
  python tools/isa_cost.py --check    # exit 1 if the committed file differs from what the sources give now
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "pathtracer_amd", "isa_cost.json")
KERNELS = {"k_extend": "_Z8k_extendILb0ELi0EE", "k_connect": "_Z9k_connectILb0ELi0EE",
           # the same kernels over the quantised 64-byte node records (trees beyond L2)
           "k_extend_quant": "_Z8k_extendILb0ELi1EE", "k_connect_quant": "_Z9k_connectILb0ELi1EE"}


def assembly():
    hipcc = "hipcc" if subprocess.run(["which", "hipcc"], capture_output=True).returncode == 0 else "/opt/rocm/bin/hipcc"
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "marks.s")
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-DPT_ISA_MARKS", "--cuda-device-only", "-S",
               "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "pathtracer_amd", "csrc"), "-o", out,
               os.path.join(ROOT, "pathtracer_amd", "csrc", "pt_api.hip")]
        subprocess.run(cmd, check=True, capture_output=True)
        return open(out).read().splitlines()


def function_lines(lines, prefix):
    start = next(i for i, l in enumerate(lines) if l.startswith(prefix) and ":" in l.split(";")[0])   # the label line "<mangled name>:"
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
    return lines[start + 1:end]


def blocks_of(fn):
    """Basic blocks: split at labels, after branches and at markers.  Returns (blocks, label -> block index)."""
    blocks, labels = [[]], {}
    for raw in fn:
        line = raw.split(";")[0].strip() if "PT_MARK" not in raw else raw.strip()
        if not line:
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m:
            if blocks[-1]:
                blocks.append([])
            labels[m.group(1)] = len(blocks) - 1
            continue
        if line.startswith("."):
            continue
        if line.startswith("; PT_MARK"):
            if blocks[-1]:
                blocks.append([])
            blocks[-1].append(line)
            blocks.append([])
            continue
        blocks[-1].append(line)
        if line.startswith("s_cbranch") or line.startswith("s_branch") or line.startswith("s_endpgm") or line.startswith("s_setpc"):
            blocks.append([])
    return blocks, labels


def successors(blocks, labels):
    succ = [[] for _ in blocks]
    for i, b in enumerate(blocks):
        last = b[-1] if b else ""
        tgt = re.search(r"(\.LBB\d+_\d+)", last)
        if last.startswith("s_branch"):
            succ[i].append(labels[tgt.group(1)])
        elif last.startswith("s_endpgm") or last.startswith("s_setpc"):
            pass
        else:
            if last.startswith("s_cbranch"):
                succ[i].append(labels[tgt.group(1)])
            if i + 1 < len(blocks):
                succ[i].append(i + 1)
    return succ


def reach(start, stop, edges):
    seen, work = set(), [start]
    while work:
        n = work.pop()
        for m in edges[n]:
            if m != stop and m not in seen:
                seen.add(m)
                work.append(m)
    return seen


def classify(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "scratch_", "buffer_")):
        return "vmem"
    if op.startswith("s_"):
        return "salu"
    return "other"


# Instructions measured at about one per 2.5 clocks per SIMD with several waves resident (tools/micro/valu_rate.hip,
# profiles/r03_micro_valu_rate.txt), in their 32-bit encoding; every other vector-ALU instruction issues at about one per 4.2
FAST = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_add_u32", "v_sub_u32", "v_subrev_u32",
        "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_mov_b32", "v_ashrrev_i32"}


def is_fast(op):
    return op.endswith("_e32") and op[:-4] in FAST


def body_cost(blocks, succ, name):
    begin = next(i for i, b in enumerate(blocks) if b and b[0] == f"; PT_MARK {name}_begin")
    end = next(i for i, b in enumerate(blocks) if b and b[0] == f"; PT_MARK {name}_end")
    pred = [[] for _ in blocks]
    for i, ss in enumerate(succ):
        for s in ss:
            pred[s].append(i)
    body = reach(begin, end, succ) & reach(end, begin, pred)
    # a rarely executed part of the body (bracketed by cold_begin / cold_end markers) does not belong to the cost of an iteration
    cold_b = [i for i in body if blocks[i] and blocks[i][0] == "; PT_MARK cold_begin"]
    cold_e = [i for i in body if blocks[i] and blocks[i][0] == "; PT_MARK cold_end"]
    cold = set()
    for b in cold_b:
        for e in cold_e:
            cold |= reach(b, e, succ) & reach(e, b, pred)
    body -= cold
    cost = {"valu": 0, "valu_fast": 0, "salu": 0, "vmem": 0, "lds": 0, "other": 0}
    for i in sorted(body):
        for ins in blocks[i]:
            op = ins.split()[0]
            cost[classify(op)] += 1
            if is_fast(op):
                cost["valu_fast"] += 1     # of the "valu" count
    cost["blocks"] = len(body)
    cost["cold_blocks_left_out"] = len(cold)
    return cost


def measure():
    lines = assembly()
    out = {}
    for kernel, prefix in KERNELS.items():
        blocks, labels = blocks_of(function_lines(lines, prefix))
        succ = successors(blocks, labels)
        out[kernel] = {"interior_iteration": body_cost(blocks, succ, "interior"), "leaf_phase": body_cost(blocks, succ, "leaf")}
    out["method"] = ("instructions of every basic block on a path between the PT_MARK begin/end comments of the -DPT_ISA_MARKS build "
                     "(hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -S); inner loops counted once; tools/isa_cost.py")
    return out


def main():
    got = measure()
    if "--check" in sys.argv:
        have = json.load(open(OUT)) if os.path.exists(OUT) else None
        if have != got:
            print("pathtracer_amd/isa_cost.json is stale: run python tools/isa_cost.py")
            print(json.dumps(got, indent=1))
            return 1
        return 0
    json.dump(got, open(OUT, "w"), indent=1)
    print(json.dumps(got, indent=1))
    return 0


if __name__ == "__main__":
    sys.exit(main())
