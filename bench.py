#!/usr/bin/env python3
"""Benchmark of the wavefront path-tracing hot path on MI355X.

One step = one frame: `spp` samples per pixel of the workload through
generate -> {extend -> shade -> connect} x bounces on every rank's tiles (scene and rays resident
in HBM; the device-to-host image read is outside the timed region).  Default workload is
the headline configuration: the reference's bundled scene, 1920x1080, 16 spp, 4 bounces, NEE.

  python bench.py --gpus N --steps K --warmup W

starts by itself: with N > 1 and no launcher environment it spawns its N ranks through
torch.distributed.run as a child process (before anything touches the GPU) and relays the JSON line; started by
torch.distributed.run (RANK / WORLD_SIZE set) it is one of the ranks.

With N > 1 the image's 8x8 pixel groups are sharded round-robin over the ranks (scene replicated) and once per
frame the rank-local accumulators are gathered to rank 0 with RCCL inside the timed region.
  --scaling weak    (default) every rank traces spp*N samples of its groups: fixed rays per GPU
  --scaling strong  the frame is fixed (spp samples of every pixel), each rank traces its 1/N of the pixels
Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (description, width, height, spp, bounces, nee)
    "bundled": ("bundled scene (Game.InitializeScene(11), 165,208 triangles), 1920x1080, 16 spp, 4 bounces, NEE", 1920, 1080, 16, 4, 1),
    "soup1m": ("synthetic 1M-triangle random soup + SAH BVH, 1920x1080, 8 spp, 4 bounces, diffuse only, NEE", 1920, 1080, 8, 4, 1),
    # the other benchmark configurations (parity/measurement cases, not the default bench line)
    "c1": ("bundled scene, 512x512, 1 spp, primary + 1 bounce, no NEE (the reference's own setting)", 512, 512, 1, 2, 0),
    "c3": ("synthetic 1M-triangle random soup + SAH BVH, 1920x1080, 64 spp, 4 bounces, diffuse only, NEE", 1920, 1080, 64, 4, 1),
    "c4": ("bundled recipe + 19 bunnies (259,600 triangles, Sponza-scale), 3840x2160, 256 spp in steps of 32, 4 bounces, NEE", 3840, 2160, 32, 4, 1),
    "c5": ("4,026 instanced bunnies flattened (20.0M triangles), 3840x2160, 1024 spp in steps of 32, 8 bounces, NEE", 3840, 2160, 32, 8, 1),
}

# Vector memory.  A CU's texture path returns 64 bytes per clock when the lanes of a wave instruction cover whole cache lines
# (tools/micro/gather_rate.hip: 1,152-byte rows gathered by whole waves read at 115-127 GB/s per CU = 30-32 TB/s from L2,
# profiles/r03_micro_gather_rate*.txt): 256 CUs x 64 B x 2.4 GHz is the roof.  The traversal's own pattern — every lane fetches the
# record of its own ray, 64 different lines per wave instruction — is served at one lane-load (16 B) per clock and CU,
# 9.8 TB/s nominal and 9.5-9.7 measured from L2, 4-5 from the Infinity Cache: that ceiling is reported beside the roof.
VMEM_PEAK_GBS = 256 * 64 * 2.4e9 / 1e9
VMEM_PER_LANE_PATTERN_GBS = 256 * 16 * 2.4e9 / 1e9
# Lane-operations the reference's own arithmetic needs, whatever the kernel's instruction count: a step over four boxes is four slab
# tests (Box.cs:51-79: 6 subtractions, 6 multiplications, 10 min/max = 22 each, unfused) plus 12 to order the survivors (five
# compare-exchanges and the choice of the nearest); a triangle test is Moeller-Trumbore as Triangle.cs:15-47 writes it: two cross
# products (18), four dot products (20), a reciprocal, a subtraction of points (3), three scalings and eight comparisons = 53.
ALG_LANEOPS_PER_STEP = 4 * 22 + 12
ALG_LANEOPS_PER_TRI = 53
# Vector-ALU peak, second entry: one wave64 instruction per SIMD every 4 clocks, i.e. 16 lanes per
# SIMD and clock, 4 SIMDs x 256 CUs at 2.4 GHz.  Not the 157 TFLOP/s headline: that counts packed two-wide FMAs, and this path
# may neither fuse (bit-exactness with the C# arithmetic) nor pack (one ray per lane).
VALU_PEAK_TLANEOPS = 256 * 4 * 16 * 2.4e9 / 1e12
# What the SIMDs were measured to issue with several waves resident (tools/micro/valu_rate.hip, profiles/r03_micro_valu_rate.txt):
# float add / sub / mul / fma and the plain two-operand integer and logic instructions about one wave64 instruction per 2.5 clocks,
# min / max / selects / compares / three-operand instructions one per 4.2 (clocks of the nominal 2.4 GHz).  isa_cost.json counts the
# first kind apart (valu_fast), and the line reports the share of a launch's SIMD time those rates account for (issue_time_frac).
VALU_CLOCKS_FAST, VALU_CLOCKS_FULL = 2.5, 4.2
HBM_PEAK_GBS = 8000.0


def host_threads():
    """Threads for the CPU baseline: the CPUs this job may use (a 1-GPU box grants a share of the machine)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    share = os.environ.get("OMP_NUM_THREADS")
    if share and share.isdigit():
        n = min(n, int(share))
    return max(1, n)


def build_scene(name, threads):
    from pathtracer_amd.host import Scene
    if name in ("bundled", "c1"):
        s = Scene()
        s.initialize_scene(11)
    elif name == "c4":
        s = Scene()
        s.initialize_scene_extended(11, 19)
    elif name == "c5":
        s = Scene()
        s.generate_instances(4026, 11)
    else:
        s = Scene("")
        s.generate_soup(1000000)
    s.build(threads)
    return s


def view_of(name):
    if name in ("bundled", "c1", "c4"):
        # Game.Initialize sets the camera before the scene exists: focal distance stays 20 (Game.cs:45-51)
        return (0, 2, 15), (0, 3, 0)
    if name == "c5":
        return (0, 25, 70), (0, 18, 0)
    return (0, 0, 120), (0, 0, 0)


def count_quads(nodes):
    """Interior nodes at an even depth = quad records of the device tree (one per two levels)."""
    if nodes.size == 0 or nodes["count"][0] > 0:
        return 0
    level = np.array([0], np.int64)
    depth, quads = 0, 0
    while level.size:
        interior = level[nodes["count"][level] <= 0]
        if depth % 2 == 0:
            quads += int(interior.size)
        left = nodes["left_first"][interior].astype(np.int64)
        level = np.concatenate([left, left + 1])
        depth += 1
    return quads


def cpu_baseline(name, w, h, bounces, nee, fs, threads):
    """The oracle's literal restatement of the C# CPU traversal + shading on a bounded sample of the workload."""
    from oracle import binding as ob
    if name in ("bundled", "c1"):
        o = ob.Scene(os.path.join(ROOT, "pathtracer_amd", "assets", "meshes.bin"))
        o.initialize_scene(11)
    else:
        o = ob.Scene("")
        o.soup(1000000)
    o.build()
    rs = ob.render_state(w, h)
    cfg = ob.path_config(bounces, nee, 1)
    spp = 4      # of the frame's 16: the same pixels, cameras and seeds as the GPU's first 4 samples
    t0 = time.perf_counter()
    osum, st = o.render(rs, fs, cfg, spp, traversal="literal", threads=threads)
    dt = time.perf_counter() - t0
    rays = st["extend_rays"] + st["shadow_rays"]
    stride = 4   # the reference is single-threaded: one thread on every 4th 8x8 pixel group, 1 spp
    t1 = time.perf_counter()
    _, st1 = o.render(rs, fs, cfg, 1, traversal="literal", threads=1, group_stride=stride)
    dt1 = time.perf_counter() - t1
    rays1 = st1["extend_rays"] + st1["shadow_rays"]
    # the image for the parity check: the same samples through the oracle's canonical walk (closest hit with ties to the smaller
    # triangle index — the contract of the GPU path); the literal order may keep another triangle of an exact distance tie
    csum, _ = o.render(rs, fs, cfg, spp, traversal="canonical", threads=threads)
    literal_ties = int((csum.view(np.uint32) != osum.view(np.uint32)).any(axis=-1).sum())
    return csum / np.float32(spp), literal_ties, {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "host_cpu_count": os.cpu_count(),
            "sample": f"{spp} of the frame's samples per pixel, whole image ({rays} rays in {dt:.1f} s on {threads} threads = every CPU this "
                      f"job may use; os.cpu_count() = {os.cpu_count()} is the whole machine); "
                      f"C++ -O2 restatement of Accelerator.Intersect + shade, which is faster than the C# original",
            "single_thread_value": rays1 / dt1 / 1e6,
            "single_thread_sample": f"1 spp of every {stride}th pixel group ({rays1} rays in {dt1:.1f} s on 1 thread)"}


def isa_cost():
    """VALU instructions per interior iteration / leaf phase of the traversal kernels (tools/isa_cost.py, from the kernels' ISA)."""
    with open(os.path.join(ROOT, "pathtracer_amd", "isa_cost.json")) as f:
        return json.load(f)


def committed_pmc(workload):
    """HBM-side traffic and VALU-busy of the extend kernel from the PMC passes committed under profiles/ (NOT measured by this run)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        return json.load(open(path)).get(workload, {})
    except Exception:
        return {}


class Run:
    """One workload on this rank's context: scene, camera, step()."""

    def __init__(self, args, name, rank, world, local_rank, spp_override=0, strong=False):
        import pathtracer_amd as pa
        from pathtracer_amd import tiles
        from pathtracer_amd.host import Camera
        self.name = name
        self.desc, w, h, spp, bounces, nee = WORKLOADS[name]
        self.w, self.h = args.width or w, args.height or h
        self.spp, self.bounces, self.nee = spp_override or args.spp or spp, args.bounces or bounces, nee
        self.world, self.rank = world, rank
        self.spp_total = self.spp if strong else self.spp * world
        scene = build_scene(name, threads=min(16, host_threads()))
        self.mats, self.tris, self.nodes = scene.arrays()
        self.n_quads = count_quads(self.nodes)
        # the library's rule: trees whose exact records exceed the 32 MiB of L2 are walked through the quantised 64-byte records
        self.quant = self.n_quads * 128 > (32 << 20) if args.quant == 2 else bool(args.quant)
        self.in_process = getattr(args, "in_process", 0)
        if self.in_process:
            # one process, one context over N GPUs: tiles and the RCCL gather are inside the library (pt_config.device_count)
            self.wf = pa.Wavefront(device=0, nee=bool(nee), rng_mode=1, samples_per_batch=args.batch, stage_timing=1, device_count=self.in_process,
                                   group_flags=1 if args.same_device else 0)
            self.spp_total = self.spp if strong else self.spp * self.in_process
        else:
            self.wf = pa.Wavefront(device=local_rank, nee=bool(nee), rng_mode=1, tile_rank=rank, tile_world=world,
                                   samples_per_batch=args.batch, stage_timing=1)
        if getattr(args, "no_pipeline", False):
            self.wf.set_option(11, 0)
        self.instanced = bool(getattr(args, "instanced", 0))
        if self.instanced:
            # the same scene, built on the GPU, with the tree over instances beside the flattened one; the kernels walk the instances
            mats, refs = scene.references()
            self.wf.build_scene_instanced(mats, refs, *scene.instancing())
            self.wf.set_option(10, 1)
            self.quant = False
        else:
            self.wf.upload_scene(self.mats, self.tris, self.nodes)
        if args.blocks >= 0:
            self.wf.set_option(6, args.blocks)
        if args.refill_primary:
            self.wf.set_option(7, args.refill_primary)
        if args.refill_bounce:
            self.wf.set_option(8, args.refill_bounce)
        if args.quant != 2:
            self.wf.set_option(9, args.quant)
        self.wf.allocate(self.w, self.h)
        self.pos, self.tgt = view_of(name)
        self.cam = Camera(self.w, self.h, None)
        self.fs = self.cam.frame_state(self.pos, self.tgt, frames=1, samples=0)
        self.gather = tiles.AccumulatorGather(self.wf, dst=0) if (world > 1 and not self.in_process) else None
        self.image = None

    def step(self):
        self.wf.invoke(self.fs, spp=self.spp_total, max_bounces=self.bounces, wait=False)
        self.fs["frames"] += self.spp_total
        self.fs["samples"] += self.spp_total
        if self.gather is not None:
            self.gather()   # the image's tiles to rank 0; RCCL moves them while the next image renders
        if self.in_process:
            self.image = self.wf.read_image()   # the library's own gather (grouped ncclSend / ncclRecv) + assembly + read-out, inside the timed region

    def instrumented(self):
        """One un-timed step with the counting kernels (BVH steps, triangle tests, wave iterations), one with every stage
        launch bracketed by events (the stage breakdown)."""
        wf = self.wf
        wf.set_option(3, 1)
        wf.reset_stats()
        self.step()
        wf.synchronize()
        counted = wf.stats()
        wf.set_option(3, 0)
        wf.set_option(4, 1)
        wf.reset_stats()
        self.step()
        wf.synchronize()
        staged = wf.stats()
        return counted, staged

    def close(self):
        self.wf.close()


def traversal_roofline(counted, timed, kernel_cost, n_quads, n_tris, pmc, quant=False, gpus=1):
    """The extend kernel against the roof that binds it — the vector memory pipeline's return rate — from counts of the
    instrumented pass and HIP-event times of the timed steps.

    Algorithmic bytes per launch = 16 B x the DISTINCT 16-byte loads the walk needs: 7 per quad record a wave instruction fetches
    (four boxes + four entries = 112 bytes), 3 per triangle record (48 bytes), 2 per ray (origin, direction); lanes of one wave
    instruction that fetch the same record count once (the instrumented kernel counts distinct records per instruction).
    Second and third entries: vector-ALU issue (useful lane-operations from the ISA's instruction counts) and HBM."""
    i_int, i_leaf = kernel_cost["interior_iteration"]["valu"], kernel_cost["leaf_phase"]["valu"]
    rays = max(1, counted["extend_rays"])
    steps_per_ray = counted["extend_steps"] / rays
    tris_per_ray = counted["extend_tris"] / rays
    usteps_per_ray = counted["extend_unique_steps"] / rays
    utris_per_ray = counted["extend_unique_tris"] / rays
    launches = max(1, timed["extend_launches"])
    avg_ms = timed["ms_extend"] / launches
    sec = max(1e-12, avg_ms * 1e-3)
    rays_per_launch = timed["extend_rays"] / launches / gpus   # a multi-GPU context adds the GPUs' rays up and reports launches per GPU
    # 16-byte loads per record: exact quad record 7 (112 of 128 bytes), triangle 3; the quantised walk's records are 4 and 4
    step_loads, tri_loads = (4.0, 4.0) if quant else (7.0, 3.0)
    loads_per_ray = step_loads * usteps_per_ray + tri_loads * utris_per_ray + 2.0
    vmem_achieved = 16.0 * loads_per_ray * rays_per_launch / sec / 1e9
    laneops_per_ray = steps_per_ray * i_int + tris_per_ray * i_leaf
    valu_achieved = laneops_per_ray * rays_per_launch / sec / 1e12
    alg_laneops_per_ray = steps_per_ray * ALG_LANEOPS_PER_STEP + tris_per_ray * ALG_LANEOPS_PER_TRI
    valu_alg_achieved = alg_laneops_per_ray * rays_per_launch / sec / 1e12
    lane_int = counted["extend_steps"] / max(1, 64 * counted["sched_interior"])
    lane_leaf = counted["extend_tris"] / max(1, 64 * counted["sched_leaf"])
    issued = counted["sched_interior"] * i_int + counted["sched_leaf"] * i_leaf     # wave instructions, the two loops only
    # the same instructions priced at the rates the SIMDs were measured to issue them at: the share of the launches' SIMD time they take
    f_int, f_leaf = kernel_cost["interior_iteration"].get("valu_fast", 0), kernel_cost["leaf_phase"].get("valu_fast", 0)
    issue_clocks = (counted["sched_interior"] * (f_int * VALU_CLOCKS_FAST + (i_int - f_int) * VALU_CLOCKS_FULL) +
                    counted["sched_leaf"] * (f_leaf * VALU_CLOCKS_FAST + (i_leaf - f_leaf) * VALU_CLOCKS_FULL))
    issue_time_frac = issue_clocks / max(1, counted["extend_launches"]) / (sec * 2.4e9 * 256 * 4 * gpus)
    # HBM: the bytes a launch cannot avoid moving are its ray records (32 in, 16 out) and one pass over the tree and the hot
    # triangle records; everything else the walk touches is re-use that the caches serve
    scene_bytes = (64.0 * n_quads + 64.0 * n_tris) if quant else (128.0 * n_quads + 48.0 * n_tris)
    hbm_alg = 48.0 * rays_per_launch + scene_bytes
    hbm_achieved = hbm_alg / sec / 1e9
    vmem = {"bound": "vmem", "bound_note": "vector memory: 64 B per clock and CU for wave-coalesced lines (tools/micro/gather_rate.hip, profiles/r03_micro_gather_rate.txt); "
                                           "the per-lane record fetch of a traversal step is served at a quarter of that (frac_of_per_lane_pattern_ceiling)",
            "achieved": vmem_achieved, "peak": VMEM_PEAK_GBS, "unit": "GB/s", "frac": vmem_achieved / VMEM_PEAK_GBS,
            "frac_of_per_lane_pattern_ceiling": vmem_achieved / VMEM_PER_LANE_PATTERN_GBS, "per_lane_pattern_ceiling": VMEM_PER_LANE_PATTERN_GBS,
            "bytes_per_launch": 16.0 * loads_per_ray * rays_per_launch, "distinct_16B_loads_per_ray": loads_per_ray,
            "distinct_bvh_records_per_ray": usteps_per_ray, "distinct_tri_records_per_ray": utris_per_ray,
            "node_records": "quantised, 64 B" if quant else "exact, 128 B",
            "measured_ceilings_gbs": {"source": "profiles/r03_micro_gather_rate.txt, profiles/r03_micro_coop_fetch.txt", "wave_coalesced_rows_from_L2": 31600.0,
                                      "per_lane_records_from_L2": 9700.0, "per_lane_records_from_infinity_cache": 7300.0}}
    valu = {"bound": "valu", "bound_note": "vector-ALU issue: one wave64 instruction per SIMD every 4 clocks, unfused and unpacked (bit-exactness with the C# "
                                           "arithmetic forbids FMA contraction); binds coherent rays, whose loads the pipeline serves together",
            "achieved": valu_achieved, "peak": VALU_PEAK_TLANEOPS, "unit": "Tlane-op/s", "frac": valu_achieved / VALU_PEAK_TLANEOPS,
            # the same with the reference arithmetic's own operation counts instead of the kernel's instruction counts
            "achieved_algorithmic": valu_alg_achieved, "frac_algorithmic": valu_alg_achieved / VALU_PEAK_TLANEOPS,
            "algorithmic_lane_ops": {"per_4_box_step": ALG_LANEOPS_PER_STEP, "per_triangle_test": ALG_LANEOPS_PER_TRI, "per_ray": alg_laneops_per_ray},
            "valu_instr_per_interior_iteration": i_int, "valu_instr_per_leaf_phase": i_leaf, "lane_ops_per_ray": laneops_per_ray,
            "lane_utilisation": {"interior": lane_int, "leaf": lane_leaf,
                                 "overall": (counted["extend_steps"] * i_int + counted["extend_tris"] * i_leaf) / max(1, 64 * issued)},
            "issue_model_wave_instr_per_ray": issued / rays, "valu_busy_pmc": pmc.get("valu_busy"),
            "issue_time_frac": issue_time_frac, "valu_fast_instr": {"interior_iteration": f_int, "leaf_phase": f_leaf},
            "issue_time_note": "wave iterations x (fast instructions x 2.5 + the others x 4.2 clocks) over the launches' SIMD clocks: float add/sub/mul/fma and plain "
                               "two-operand integer instructions were measured at one per 2.5 clocks per SIMD, the rest at one per 4.2 (profiles/r03_micro_valu_rate.txt)",
            # cross-check of the instruction counts: wave iterations x VALU instructions per iteration against the hardware's count of
            # the same kernel (committed PMC pass; the two loops only, so a little below 1 is expected)
            "issue_model_over_sq_insts_valu": (issued / max(1, counted["extend_launches"]) / pmc["sq_insts_valu_per_launch"]) if pmc.get("sq_insts_valu_per_launch") else None}
    hbm = {"bound": "hbm", "bound_note": "ray records in and out plus one pass over the tree and the hot triangle records per launch; the walk's re-use is served by L2 / Infinity Cache",
           "achieved": hbm_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_achieved / HBM_PEAK_GBS, "bytes_per_launch": hbm_alg, "scene_bytes": scene_bytes}
    # the roof that binds is the one the kernel sits closest to; the other two stay in the line
    first, second = (vmem, valu) if vmem["frac"] >= valu["frac"] else (valu, vmem)
    out = dict(first)
    out.setdefault("frac_algorithmic", valu["frac_algorithmic"])
    out.setdefault("lane_utilisation", valu["lane_utilisation"])
    out.update({
        "kernel": "k_extend",
        "traffic": pmc.get("extend_hbm_bytes_per_launch"),
        "traffic_source": (pmc.get("source", "profiles/traffic.json") + " (committed PMC passes, not measured by this run)") if pmc else None,
        "launches": launches, "avg_launch_ms": avg_ms, "rays_per_launch": rays_per_launch,
        "bvh_steps_per_ray": steps_per_ray, "tri_tests_per_ray": tris_per_ray,
        "extend_mrays_per_s": timed["extend_rays"] / max(1e-9, timed["ms_extend"] * 1e-3) / 1e6,
        "other_roofs": {second["bound"]: second, "hbm": hbm},
    })
    return out


def timed_steps(run, steps, warmup, sync_all):
    for _ in range(warmup):
        run.step()
    sync_all()
    run.wf.reset_stats()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(steps):
        run.step()
    sync_all()
    return time.perf_counter() - t0, run.wf.stats()


def secondary_soup(args, cost, steps=3):
    """The north-star workload beside the headline line: 1M-triangle soup, 1080p, 64 spp, 4 bounces, NEE on this GPU."""
    run = Run(args_without_overrides(args), "c3", 0, 1, 0)
    counted, staged = run.instrumented()

    def sync_all():
        run.wf.synchronize()

    # the production schedule (no events; over the quantised records connect of a bounce runs inside the next bounce's extend launch),
    # as the headline steps are timed; then one frame with events around the extend launches, which also gives every stage launches of
    # its own (connect beside the next extend on a second stream), for the launch durations the roofline divides by
    run.wf.set_option(4, 0)
    dt, st_run = timed_steps(run, steps, 1, sync_all)
    run.wf.set_option(4, 2)
    _, st = timed_steps(run, 1, 0, sync_all)
    # per-stage figures: from the one frame whose every stage was bracketed by events, stages back to back on one stream (in the timed
    # frames connect of a bounce runs beside extend of the next one, and an extend launch's events then span shared time)
    bounce_rays = staged["extend_rays"] - staged["paths"]
    bounce_ms = staged["ms_extend"] - staged["ms_extend_primary"]
    kcost = cost["k_extend_quant" if run.quant else "k_extend"]
    # live: the extend launches of the timed frames (a launch may share the GPU with the previous bounce's connect for part of its
    # time) — what a kernel trace of this command shows; alone: the frame whose stages ran back to back
    roof = traversal_roofline(counted, st, kcost, run.n_quads, int(run.tris.size), committed_pmc("c3"), run.quant)
    alone = traversal_roofline(counted, staged, kcost, run.n_quads, int(run.tris.size), committed_pmc("c3"), run.quant)
    out = {
        "workload": run.desc, "steps": steps, "ms_per_frame": dt / steps * 1e3,
        "mrays_per_s": (st_run["extend_rays"] + st_run["shadow_rays"]) / dt / 1e6,
        "extend_mrays_per_s": staged["extend_rays"] / max(1e-9, staged["ms_extend"] * 1e-3) / 1e6,
        "bounce_extend_mrays_per_s": bounce_rays / max(1e-9, bounce_ms * 1e-3) / 1e6,
        "primary_extend_mrays_per_s": staged["paths"] / max(1e-9, staged["ms_extend_primary"] * 1e-3) / 1e6,
        "connect_mrays_per_s": staged["shadow_rays"] / max(1e-9, staged["ms_connect"] * 1e-3) / 1e6,
        "stage_ms_per_frame": {k: staged["ms_" + k] for k in ("generate", "extend", "shade", "connect")},
        "stage_ms_note": "stages back to back on one stream (one frame); ms_per_frame is the production schedule, connect of a bounce inside the next bounce's extend launch; "
                         "roofline: one more frame with events around its extend launches and connect beside the next extend on a second stream",
        "rays_per_frame": {"extend": st_run["extend_rays"] / steps, "bounce": bounce_rays, "shadow": st_run["shadow_rays"] / steps},
        "shadow_bvh_steps_per_ray": counted["shadow_steps"] / max(1, counted["shadow_rays"]),
        "shadow_tri_tests_per_ray": counted["shadow_tris"] / max(1, counted["shadow_rays"]),
        "node_records": "quantised, 64 B" if run.quant else "exact, 128 B",
        "roofline": {**{k: roof.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_algorithmic", "avg_launch_ms", "launches", "rays_per_launch",
                                                 "bvh_steps_per_ray", "tri_tests_per_ray", "lane_utilisation")},
                     "other_roofs": {k: {"frac": v["frac"], "achieved": v["achieved"], "peak": v["peak"], "unit": v["unit"]} for k, v in roof["other_roofs"].items()},
                     "kernel_alone": {k: alone.get(k) for k in ("bound", "achieved", "frac", "frac_algorithmic", "avg_launch_ms")}},
    }
    run.close()
    return out


def args_without_overrides(args):
    a = argparse.Namespace(**vars(args))
    a.width = a.height = a.spp = a.bounces = 0
    return a


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="bundled", choices=sorted(WORKLOADS))
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--bounces", type=int, default=0)
    ap.add_argument("--batch", type=int, default=0, help="samples traced concurrently (0 = automatic)")
    ap.add_argument("--blocks", type=int, default=-1, help="resident traversal blocks per CU (-1 = the library's choice, 0 = as many as fit)")
    ap.add_argument("--refill-primary", type=int, default=0, help="idle lanes a traversal wave collects before fetching camera rays (0 = the library's choice)")
    ap.add_argument("--refill-bounce", type=int, default=0, help="the same for bounce and shadow rays")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the 1M-triangle soup block of the N = 1 line")
    ap.add_argument("--backend", default="nccl", help="process-group backend; gloo only for rehearsing N > 1 on a single GPU")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses device 0")
    ap.add_argument("--quant", type=int, default=2, choices=[0, 1, 2], help="node records the traversal walks: 0 exact 128-byte, 1 quantised 64-byte, 2 the library's rule")
    ap.add_argument("--instanced", action="store_true", help="walk the scene as instances of meshes (pt_build_scene_instanced + PT_OPT_INSTANCED): opt-in mode, "
                    "same hit distances, see pt_hip.h")
    ap.add_argument("--no-pipeline", action="store_true", help="one set of streams and queues (PT_OPT_PIPELINE_CALLS = 0): consecutive frames do not overlap")
    ap.add_argument("--in-process", action="store_true", help="N GPUs behind ONE context in this one process (pt_config.device_count = N: tile sharding and the "
                    "RCCL gather inside libpt_hip.so) instead of one process per GPU")
    args = ap.parse_args()

    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.same_device and args.gpus > 1 and args.backend == "nccl" and not args.in_process:
        # RCCL refuses two ranks of one communicator on one device (ncclInvalidUsage, "Duplicate GPU detected"), and some versions wait
        # in the bootstrap instead of returning: say so instead of starting ranks that cannot finish
        print("bench.py: --same-device with --backend nccl cannot work: RCCL needs one GPU per rank (ncclCommInitRank fails with "
              "ncclInvalidUsage 'Duplicate GPU detected' when two ranks name one device). Rehearse with --backend gloo, or use --in-process "
              "--same-device (members of one context share the GPU and move tiles by device copies).", file=sys.stderr, flush=True)
        return 2
    args.in_process = args.gpus if args.in_process else 0
    if args.in_process:
        launched = False          # one process drives all the GPUs
    if not launched and args.gpus > 1 and not args.in_process:
        # Start the N ranks ourselves, as fresh child processes, before this process has touched the GPU (it never will).
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        return subprocess.run(cmd, env=env).returncode

    rank = int(os.environ.get("RANK", "0")) if not args.in_process else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if not args.in_process else 0
    world = int(os.environ.get("WORLD_SIZE", "1")) if not args.in_process else 1
    if not args.in_process:
        args.gpus = world

    import torch
    import torch.distributed as dist

    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    strong = args.scaling == "strong"
    run = Run(args, args.workload, rank, world, local_rank, strong=strong)
    wf = run.wf

    def sync_all():
        wf.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    counted, staged = run.instrumented()
    stage_ms = {k: staged["ms_" + k] for k in ("generate", "extend", "shade", "connect")}
    # the timed steps bracket only the extend launches (every pair of events costs stream time), and not even those when the launch
    # duration is taken from the frames rendered one at a time afterwards
    pipelined = not getattr(args, "no_pipeline", False)
    wf.set_option(4, 0 if pipelined else 2)
    dt, st = timed_steps(run, args.steps, args.warmup, sync_all)
    # The timed steps run two frames in flight (consecutive pt_render calls overlap): an extend launch then shares the GPU with the
    # other frame's kernels, and the span between the events around it includes the time its blocks wait to be placed.  The kernel's
    # own launch duration — what the roofline divides by, and what rocprofv3's kernel statistics of this command show — is taken from
    # a few more frames with one frame in flight (connect of a bounce still beside extend of the next, as a lone frame runs).
    st_span = st
    if pipelined:
        wf.set_option(11, 0)
        wf.set_option(4, 2)
        _, st = timed_steps(run, min(args.steps, 3), 0, sync_all)
        wf.set_option(11, 1)

    red_dev = "cuda" if (args.backend == "nccl" or world == 1) else "cpu"
    tot = torch.tensor([float(st_span["extend_rays"]), float(st_span["shadow_rays"])], dtype=torch.float64, device=red_dev)
    tmax = torch.tensor([dt], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    extend_rays, shadow_rays = tot[0].item(), tot[1].item()
    dt = tmax[0].item()

    if rank == 0:
        cost = isa_cost()
        kcost = cost["k_extend_quant" if run.quant else "k_extend"]
        roof = traversal_roofline(counted, st, kcost, run.n_quads, int(run.tris.size), committed_pmc(args.workload), run.quant, args.in_process or 1)
        roof["shadow_bvh_steps_per_ray"] = counted["shadow_steps"] / max(1, counted["shadow_rays"])
        roof["shadow_tri_tests_per_ray"] = counted["shadow_tris"] / max(1, counted["shadow_rays"])
        # in the timed steps connect of a bounce runs beside extend of the next one: an extend launch shares the GPU for part of its
        # time.  The same figures from the frame whose stages ran back to back, the kernel having the GPU to itself:
        alone = traversal_roofline(counted, staged, kcost, run.n_quads, int(run.tris.size), committed_pmc(args.workload), run.quant, args.in_process or 1)
        roof["kernel_alone"] = {k: alone[k] for k in ("bound", "achieved", "frac", "frac_algorithmic", "avg_launch_ms", "extend_mrays_per_s")}
        roof["kernel_alone"]["fracs"] = {alone["bound"]: alone["frac"], **{k: v["frac"] for k, v in alone["other_roofs"].items()}}
        roof["kernel_alone"]["bounce_extend_mrays_per_s"] = ((staged["extend_rays"] - staged["paths"]) /
                                                             max(1e-9, (staged["ms_extend"] - staged["ms_extend_primary"]) * 1e-3) / 1e6)
        bounce_rays = st["extend_rays"] - st["paths"]
        bounce_ms = st["ms_extend"] - st["ms_extend_primary"]
        roof["bounce_extend_mrays_per_s"] = bounce_rays / max(1e-9, bounce_ms * 1e-3) / 1e6
        roof["primary_extend_mrays_per_s"] = st["paths"] / max(1e-9, st["ms_extend_primary"] * 1e-3) / 1e6
        if st_span is not st:
            roof["launch_timing"] = ("HIP events around the extend launches of frames rendered one at a time after the timed steps; in the timed steps two frames are "
                                     "in flight, a launch shares the GPU with the other frame's kernels, and no events are recorded")
        overridden = args.spp or args.width or args.height or args.bounces
        out = {
            "metric": "Mrays/s (primary+bounce+shadow rays traced)", "value": (extend_rays + shadow_rays) / dt / 1e6, "unit": "Mrays/s",
            "n_gpus": args.in_process or world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": run.desc if not overridden else f"{args.workload} {run.w}x{run.h} {run.spp} spp {run.bounces} bounces nee={run.nee}",
                       "width": run.w, "height": run.h, "spp_per_gpu": run.spp, "spp_total": run.spp_total, "max_bounces": run.bounces,
                       "nee": run.nee, "triangles": int(run.tris.size), "nodes": int(run.nodes.size),
                       "tiles": (f"8x8 groups round-robin over {args.in_process} GPUs behind one context (pt_config.device_count), gather + read-out inside the timed region"
                                 if args.in_process else f"8x8 groups round-robin over {world} rank(s)"),
                       "node_records": "instances of meshes (two-level), exact 128 B" if run.instanced else ("quantised, 64 B" if run.quant else "exact, 128 B"),
                       "rays_per_step": (extend_rays + shadow_rays) / args.steps,
                       "extend_rays_per_step": extend_rays / args.steps, "shadow_rays_per_step": shadow_rays / args.steps},
            "roofline": roof,
            "stage_ms_per_step": stage_ms,   # from one un-timed step with every stage bracketed by events
        }
        if world == 1 and not args.in_process and not args.no_cpu_baseline and args.workload in ("bundled", "c1", "soup1m", "c3"):
            fs0 = run.cam.frame_state(run.pos, run.tgt, frames=1, samples=0)
            oimg, literal_ties, out["cpu_baseline"] = cpu_baseline(args.workload, run.w, run.h, run.bounces, run.nee, fs0, host_threads())
            # the checker at work: the oracle's 4 samples per pixel of the whole frame against the GPU's first 4, bit for bit
            wf.allocate(run.w, run.h)
            wf.invoke(fs0, spp=4, max_bounces=run.bounces)
            gimg = wf.read_image()
            differing = int((gimg.view(np.uint32) != oimg.view(np.uint32)).any(axis=-1).sum())
            out["cpu_baseline"]["parity_checked"] = differing == 0
            out["cpu_baseline"]["parity_note"] = (f"{differing} of {run.w * run.h} pixels of the GPU's first 4 samples per pixel differ bitwise from the oracle's image "
                                                  f"of the same samples; the literal C# visiting order, which is what the baseline times, gives {literal_ties} pixels "
                                                  "another triangle of an exact distance tie")
        else:
            out["cpu_baseline"] = None
        run.close()
        if world == 1 and not args.in_process and not args.no_secondary and args.workload == "bundled" and not overridden:
            out["secondary"] = secondary_soup(args, cost)
        print(json.dumps(out), flush=True)
    else:
        run.close()

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
