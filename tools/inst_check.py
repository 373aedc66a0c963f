#!/usr/bin/env python3
"""The instanced walk against the flattened one on the GPU: closest hits, occlusion flags and a frame, mismatches counted;
then the extend rate on incoherent bounce rays both ways.   python tools/inst_check.py [instances] [rays]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtracer_amd as pa
from pathtracer_amd.host import Scene, Camera


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000000
    s = Scene()
    if count > 0:
        s.generate_instances(count, 11)
        view = ((0, 25, 70), (0, 18, 0))
    else:
        s.initialize_scene(11)
        view = ((0, 2, 15), (0, 3, 0))
    mats, refs = s.references()
    mt, mf, mc, inst = s.instancing()
    print(f"{refs.size} triangles, {mf.size} meshes, {inst.size} instances", flush=True)
    wf = pa.Wavefront(nee=True, rng_mode=1, traversal_stats=bool(os.environ.get("INST_STATS")))
    t0 = time.time()
    ms = wf.build_scene_instanced(mats, refs, mt, mf, mc, inst)
    print(f"built in {ms:.1f} ms on the device, {time.time() - t0:.2f} s wall", flush=True)
    tri = refs["v1"]["position"]
    lo, hi = tri.min(0), tri.max(0)
    lo, hi = np.maximum(lo, -40), np.minimum(hi, 40)
    rng = np.random.RandomState(3)
    o = rng.uniform(lo, hi, (2 * n, 3)).astype(np.float32)
    d = rng.normal(size=(2 * n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    first = np.concatenate([o, d], 1).astype(np.float32)
    hit = wf.trace_closest(first)
    ok = np.nonzero(hit["tri"] >= 0)[0][:n]
    p = first[ok, :3] + first[ok, 3:] * hit["t"][ok, None]
    d2 = rng.normal(size=(len(ok), 3)).astype(np.float32)
    d2 /= np.linalg.norm(d2, axis=1, keepdims=True).astype(np.float32)
    rays = np.concatenate([p, d2], 1).astype(np.float32)
    tmax = rng.uniform(1, 60, len(ok)).astype(np.float32)
    res = {}
    for mode in (0, 1):
        wf.set_option(10, mode)
        h = wf.trace_closest(rays)
        occ = wf.trace_occluded(rays, tmax)
        wf.allocate(256, 144)
        fs = Camera(256, 144).frame_state(view[0], view[1], frames=1)
        wf.reset_stats()
        wf.invoke(fs, spp=4, max_bounces=8)
        img = wf.read_image()
        st = wf.stats()
        wf.reset_stats()
        ms, _ = wf.bench_extend(rays, iters=3)
        sb = wf.stats()
        if sb["extend_steps"]:
            print(f"   steps/ray {sb['extend_steps'] / 4 / len(rays):.2f}, tris/ray {sb['extend_tris'] / 4 / len(rays):.2f}, interior lane use "
                  f"{sb['extend_steps'] / max(1, 64 * sb['sched_interior']):.2f}, leaf lane use {sb['extend_tris'] / max(1, 64 * sb['sched_leaf']):.2f}")
        res[mode] = (h, occ, img, st, ms)
        print(f"mode {mode}: hit fraction {(h['tri'] >= 0).mean():.3f}, occluded {occ.mean():.3f}, extend {len(rays) / ms / 1e3:.0f} Mrays/s, "
              f"special rays {st['special_rays']}, overflows {st['stack_overflows']}, extend rays {st['extend_rays']}", flush=True)
    a, b = res[0], res[1]
    idd = int((a[0]["tri"] != b[0]["tri"]).sum())
    td = int((a[0]["t"].view(np.uint32) != b[0]["t"].view(np.uint32)).sum())
    od = int((a[1] != b[1]).sum())
    px = int((a[2].view(np.uint32) != b[2].view(np.uint32)).any(axis=-1).sum())
    rel = float(np.sqrt(((a[2][..., :3] - b[2][..., :3]) ** 2).mean()) / max(1e-12, np.sqrt((a[2][..., :3] ** 2).mean())))
    print(f"instanced against flattened: {idd} of {len(rays)} ids differ, {td} distances, {od} occlusion flags; {px} of {256 * 144} pixels differ, relRMSE {rel:.2e}")
    wf.close()


if __name__ == "__main__":
    main()
