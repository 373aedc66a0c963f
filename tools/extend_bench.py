#!/usr/bin/env python3
"""Extend-stage microbenchmark: incoherent bounce rays against a scene, extend kernel alone.

  python tools/extend_bench.py [--scene soup1m|bundled] [--rays 4000000] [--iters 5] [--stats]

Rays: random rays through the scene volume are traced once; the surviving hit points become
origins of uniformly random directions (the distribution of diffuse bounce rays).  Prints
Mrays/s, node-pair steps and triangle tests per ray, and a checksum of the hits so that kernel
variants can be compared for identical results.
"""
import argparse
import hashlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="soup1m")
    ap.add_argument("--rays", type=int, default=4000000)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--stats", action="store_true")
    ap.add_argument("--blocks", type=int, default=0, help="resident traversal blocks per CU (0 = as many as fit)")
    ap.add_argument("--refill", type=int, default=0, help="idle lanes a wave collects before fetching rays (0 = the library's choice for bounce rays)")
    ap.add_argument("--coherent", action="store_true", help="primary camera rays instead of bounce rays")
    ap.add_argument("--spp", type=int, default=1, help="with --coherent: this many samples of every pixel")
    ap.add_argument("--quant", type=int, default=2, help="node records: 0 exact 128-byte, 1 quantised 64-byte, 2 the library's choice (quantised for trees beyond L2)")
    ap.add_argument("--sort", type=int, default=0, help="experiment: order the rays by a Morton key of the origin with this many bits per axis (+ direction octant)")
    args = ap.parse_args()

    import pathtracer_amd as pa
    from pathtracer_amd.host import Scene, Camera

    t0 = time.time()
    if args.scene == "bundled":
        s = Scene()
        s.initialize_scene(11)
        lo, hi = np.array([-8, 0, -8], np.float32), np.array([8, 8, 8], np.float32)
    else:
        n = {"soup1m": 1000000, "soup200k": 200000, "soup4m": 4000000}[args.scene]
        s = Scene("")
        s.generate_soup(n)
        lo, hi = np.array([-50, -50, -50], np.float32), np.array([50, 50, 50], np.float32)
    s.build(threads=8)
    mats, tris, nodes = s.arrays()
    print(f"scene {args.scene}: {tris.size} triangles, {nodes.size} nodes, depth {s.counts()['depth']}, built in {time.time() - t0:.1f}s", flush=True)

    wf = pa.Wavefront(traversal_stats=args.stats)
    if args.blocks:
        wf.set_option(6, args.blocks)
    if args.refill:
        wf.set_option(8, args.refill)
    wf.upload_scene(mats, tris, nodes)
    wf.set_option(9, args.quant)
    rng = np.random.RandomState(1)
    n = args.rays

    def unit(k):
        d = rng.normal(size=(k, 3)).astype(np.float32)
        return d / np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)

    if args.coherent:
        w, h = 1920, 1080
        wf.allocate(w, h)
        pos, tgt = ((0, 2, 15), (0, 3, 0)) if args.scene == "bundled" else ((0, 0, 120), (0, 0, 0))
        fs = Camera(w, h).frame_state(pos, tgt, frames=1)
        parts = []
        for k in range(args.spp):   # sample-major, as a render call's first extend launch sees them
            fs["frames"] = 1 + k
            parts.append(wf.generate_rays(fs, w * h)[0])
        rays = np.concatenate(parts, 0)
    else:
        o = rng.uniform(lo, hi, (n * 2, 3)).astype(np.float32)
        first = np.concatenate([o, unit(n * 2)], 1)
        hit = wf.trace_closest(first)
        ok = np.nonzero(hit["tri"] >= 0)[0][:n]
        p = first[ok, :3] + first[ok, 3:] * hit["t"][ok, None]
        rays = np.concatenate([p, unit(len(ok))], 1).astype(np.float32)
    if args.sort:
        b = args.sort
        q = np.clip(((rays[:, :3] - lo) / (hi - lo) * (1 << b)).astype(np.int64), 0, (1 << b) - 1)
        key = np.zeros(rays.shape[0], np.int64)
        for bit in range(b):
            for a in range(3):
                key |= ((q[:, a] >> bit) & 1) << (3 * bit + a + 3)
        key |= (rays[:, 3] > 0).astype(np.int64) | ((rays[:, 4] > 0).astype(np.int64) << 1) | ((rays[:, 5] > 0).astype(np.int64) << 2)
        rays = np.ascontiguousarray(rays[np.argsort(key, kind="stable")])
    print(f"{rays.shape[0]} rays", flush=True)

    wf.reset_stats()
    ms, res = wf.bench_extend(rays, iters=args.iters)
    st = wf.stats()
    digest = hashlib.sha256(res["tri"].tobytes() + res["t"].tobytes()).hexdigest()[:16]
    launches = args.iters + 1
    msg = f"extend: {ms:.3f} ms/launch, {rays.shape[0] / ms / 1e3:.1f} Mrays/s, hit fraction {(res['tri'] >= 0).mean():.3f}, checksum {digest}"
    if args.stats:
        msg += f", steps/ray {st['extend_steps'] / launches / rays.shape[0]:.2f}, tris/ray {st['extend_tris'] / launches / rays.shape[0]:.2f}"
        msg += (f", interior lane use {st['extend_steps'] / max(1, 64 * st['sched_interior']):.2f}, leaf lane use {st['extend_tris'] / max(1, 64 * st['sched_leaf']):.2f}"
                f", wave iterations/64 rays: interior {st['sched_interior'] * 64 / launches / rays.shape[0]:.1f} leaf {st['sched_leaf'] * 64 / launches / rays.shape[0]:.1f} refill {st['sched_refill'] * 64 / launches / rays.shape[0]:.1f}")
    print(msg, flush=True)
    assert st["stack_overflows"] == 0
    wf.close()


if __name__ == "__main__":
    main()
