#!/usr/bin/env python3
"""Benchmark of the wavefront path-tracing hot path on MI355X.

One step = one frame: `spp` samples per pixel of the workload through
generate -> {extend -> shade -> connect} x bounces on every rank's tiles (scene and rays resident
in HBM; the device-to-host image read is outside the timed region).  Default workload is
the headline configuration: the reference's bundled scene, 1920x1080, 16 spp, 4 bounces, NEE.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

With N > 1 the image's 8x8 pixel groups are sharded round-robin over the ranks (scene replicated),
every rank traces spp*N samples of its groups (weak scaling: fixed rays per GPU), and once per
frame the rank-local accumulators are gathered to rank 0 with RCCL inside the timed region.
Rank 0 prints one JSON line.
"""
import argparse
import json
import os
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
    "c5": ("4,026 instanced bunnies flattened (20.0M triangles), 3840x2160, 1024 spp in steps of 8, 8 bounces, NEE", 3840, 2160, 8, 8, 1),
}


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


def camera_for(name, w, h, wf):
    from pathtracer_amd.host import Camera
    if name in ("bundled", "c1", "c4"):
        # Game.Initialize sets the camera before the scene exists: focal distance stays 20 (Game.cs:45-51)
        return Camera(w, h, None), (0, 2, 15), (0, 3, 0)
    if name == "c5":
        return Camera(w, h, None), (0, 25, 70), (0, 18, 0)
    return Camera(w, h, None), (0, 0, 120), (0, 0, 0)


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
    _, st = o.render(rs, fs, cfg, spp, traversal="literal", threads=threads)
    dt = time.perf_counter() - t0
    rays = st["extend_rays"] + st["shadow_rays"]
    stride = 4   # the reference is single-threaded: one thread on every 4th 8x8 pixel group, 1 spp
    t1 = time.perf_counter()
    _, st1 = o.render(rs, fs, cfg, 1, traversal="literal", threads=1, group_stride=stride)
    dt1 = time.perf_counter() - t1
    rays1 = st1["extend_rays"] + st1["shadow_rays"]
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"{spp} of the frame's samples per pixel, whole image ({rays} rays in {dt:.1f} s on {threads} threads); "
                      f"C++ -O2 restatement of Accelerator.Intersect + shade, which is faster than the C# original",
            "single_thread_value": rays1 / dt1 / 1e6,
            "single_thread_sample": f"1 spp of every {stride}th pixel group ({rays1} rays in {dt1:.1f} s on 1 thread)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="bundled", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--bounces", type=int, default=0)
    ap.add_argument("--batch", type=int, default=0, help="samples traced concurrently (0 = automatic)")
    ap.add_argument("--blocks", type=int, default=-1, help="resident traversal blocks per CU (-1 = the library's choice, 0 = as many as fit)")
    ap.add_argument("--refill-primary", type=int, default=0, help="idle lanes a traversal wave collects before fetching camera rays (0 = the library's choice)")
    ap.add_argument("--refill-bounce", type=int, default=0, help="the same for bounce and shadow rays")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="process-group backend; gloo only for rehearsing N > 1 on a single GPU")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses device 0")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N with N > 1 must be started by torch.distributed.run with N ranks")
        args.gpus = world

    import torch
    import torch.distributed as dist
    import pathtracer_amd as pa
    from pathtracer_amd import tiles

    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    desc, w, h, spp, bounces, nee = WORKLOADS[args.workload]
    w, h = args.width or w, args.height or h
    spp, bounces = args.spp or spp, args.bounces or bounces
    spp_total = spp * world

    scene = build_scene(args.workload, threads=8)
    mats, tris, nodes = scene.arrays()
    wf = pa.Wavefront(device=local_rank, nee=bool(nee), rng_mode=1, tile_rank=rank, tile_world=world,
                      samples_per_batch=args.batch, stage_timing=1)
    wf.upload_scene(mats, tris, nodes)
    if args.blocks >= 0:
        wf.set_option(6, args.blocks)
    if args.refill_primary:
        wf.set_option(7, args.refill_primary)
    if args.refill_bounce:
        wf.set_option(8, args.refill_bounce)
    wf.allocate(w, h)
    cam, pos, tgt = camera_for(args.workload, w, h, wf)
    fs = cam.frame_state(pos, tgt, frames=1, samples=0)

    gather = tiles.AccumulatorGather(wf, dst=0) if world > 1 else None

    def step():
        wf.invoke(fs, spp=spp_total, max_bounces=bounces, wait=False)
        fs["frames"] += spp_total
        fs["samples"] += spp_total
        if gather is not None:
            gather()   # the image's tiles to rank 0; RCCL moves them while the next image renders

    def sync_all():
        wf.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    # ---- un-timed instrumented pass: BVH steps and triangle tests per ray (the algorithmic bytes)
    wf.set_option(3, 1)
    wf.reset_stats()
    step()
    wf.synchronize()
    st = wf.stats()
    wf.set_option(3, 0)
    # ---- one un-timed plain pass with events around every stage launch: the stage breakdown
    wf.reset_stats()
    step()
    wf.synchronize()
    stb = wf.stats()
    stage_ms = {k: stb["ms_" + k] for k in ("generate", "extend", "shade", "connect")}
    wf.set_option(4, 2)   # the timed steps bracket only the extend launches (every pair of events costs stream time)
    steps_per_ray = st["extend_steps"] / max(1, st["extend_rays"])
    tris_per_ray = st["extend_tris"] / max(1, st["extend_rays"])
    sh_steps_per_ray = st["shadow_steps"] / max(1, st["shadow_rays"])
    sh_tris_per_ray = st["shadow_tris"] / max(1, st["shadow_rays"])

    for _ in range(args.warmup):
        step()
    sync_all()
    wf.reset_stats()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    dt = time.perf_counter() - t0

    st = wf.stats()
    red_dev = "cuda" if args.backend == "nccl" else "cpu"
    tot = torch.tensor([float(st["extend_rays"]), float(st["shadow_rays"]), dt], dtype=torch.float64, device=red_dev)
    tmax = torch.tensor([dt], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    extend_rays, shadow_rays = tot[0].item(), tot[1].item()
    dt = tmax[0].item()

    if rank == 0:
        # dominant kernel: extend.  Algorithmic bytes per ray = 32 (origin+direction in) + 16 (hit out)
        # + 128 per BVH step (one quad record: four boxes and entries) + 48 per triangle test (one hot triangle record).
        b_ray = 48.0 + 128.0 * steps_per_ray + 48.0 * tris_per_ray
        ext_launches = max(1, st["extend_launches"])
        avg_ms = st["ms_extend"] / ext_launches
        bytes_per_launch = b_ray * st["extend_rays"] / ext_launches
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic = valu_busy = None      # from the committed PMC passes of this workload (profiles/traffic.json), not measured by this run
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                pmc = json.load(open(tpath)).get(args.workload, {})
                traffic, valu_busy = pmc.get("extend_hbm_bytes_per_launch"), pmc.get("valu_busy")
            except Exception:
                traffic = valu_busy = None
        out = {
            "metric": "Mrays/s (primary+bounce+shadow rays traced)", "value": (extend_rays + shadow_rays) / dt / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc if not (args.spp or args.width or args.bounces) else f"{args.workload} {w}x{h} {spp} spp {bounces} bounces nee={nee}",
                       "width": w, "height": h, "spp_per_gpu": spp, "spp_total": spp_total, "max_bounces": bounces, "nee": nee,
                       "triangles": int(tris.size), "nodes": int(nodes.size), "tiles": f"8x8 groups round-robin over {world} rank(s)",
                       "rays_per_step": (extend_rays + shadow_rays) / args.steps,
                       "extend_rays_per_step": extend_rays / args.steps, "shadow_rays_per_step": shadow_rays / args.steps},
            "roofline": {"bound": "hbm", "kernel": "k_extend", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic, "valu_busy": valu_busy,
                         "bytes_per_ray": b_ray, "bvh_steps_per_ray": steps_per_ray, "tri_tests_per_ray": tris_per_ray,
                         "shadow_bvh_steps_per_ray": sh_steps_per_ray, "shadow_tri_tests_per_ray": sh_tris_per_ray,
                         "launches": ext_launches, "avg_launch_ms": avg_ms, "rays_per_launch": st["extend_rays"] / ext_launches,
                         "extend_mrays_per_s": st["extend_rays"] / max(1e-9, st["ms_extend"] * 1e-3) / 1e6,
                         # frac > 1 means the caches, not HBM, serve the algorithmic bytes.  What does bound a gather of random
                         # records is the rate tools/micro/tcp_coalesce.hip measures (L1/L2-resident table: 196 G records/s,
                         # Infinity-Cache-resident: 58 G records/s; profiles/r01_micro_gather.txt):
                         "gather_records_per_s": (steps_per_ray + tris_per_ray) * st["extend_rays"] / max(1e-9, st["ms_extend"] * 1e-3),
                         "gather_ceiling_records_per_s": {"l2_resident": 196e9, "infinity_cache_resident": 58e9}},
            "stage_ms_per_step": stage_ms,   # from one un-timed step with every stage bracketed by events
        }
        if world == 1 and not args.no_cpu_baseline and args.workload in ("bundled", "c1", "soup1m", "c3"):
            fs0 = cam.frame_state(pos, tgt, frames=1, samples=0)
            threads = min(16, os.cpu_count() or 1)
            out["cpu_baseline"] = cpu_baseline(args.workload, w, h, bounces, nee, fs0, threads)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)

    wf.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
