#!/usr/bin/env python3
"""GPU BVH build (pt_build_bvh) against the multi-threaded host build on the benchmark scenes.

  python tools/build_bench.py [--scene bundled|soup1m|c4|c5] [--threads 16] [--check]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="bundled")
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--check", action="store_true", help="compare the GPU arrays with a host build of the same input")
    args = ap.parse_args()
    import pathtracer_amd as pa
    from pathtracer_amd.host import Scene
    s = Scene() if args.scene != "soup1m" else Scene("")
    if args.scene == "bundled":
        s.initialize_scene(11)
    elif args.scene == "c4":
        s.initialize_scene_extended(11, 19)
    elif args.scene == "c5":
        s.generate_instances(4026, 11)
    else:
        s.generate_soup(1000000)
    t0 = time.perf_counter()
    s.build(args.threads)
    host_s = time.perf_counter() - t0
    mats, tris, nodes = s.arrays()
    print(f"scene {args.scene}: {tris.size} triangles, {nodes.size} nodes; host build on {args.threads} threads {host_s * 1e3:.0f} ms", flush=True)
    wf = pa.Wavefront()
    for it in range(3):
        t0 = time.perf_counter()
        gt, gn, ms = wf.build_bvh(tris)     # input order: the host build's output order (any order is a valid input)
        wall = time.perf_counter() - t0
        print(f"  GPU build: {ms:.1f} ms on the device, {wall * 1e3:.0f} ms with transfers and host copies; {gn.size} nodes, {tris.size / ms / 1e3:.1f} Mtris/s", flush=True)
    t0 = time.perf_counter()
    wf.upload_scene(mats, tris, nodes)
    print(f"  pt_upload_scene of host-built arrays (transfers, checks and repack on the device): {(time.perf_counter() - t0) * 1e3:.0f} ms", flush=True)
    for it in range(2):
        t0 = time.perf_counter()
        ms = wf.build_scene(mats, tris)
        print(f"  pt_build_scene (build + repack on the device): {ms:.1f} ms on the device, {(time.perf_counter() - t0) * 1e3:.0f} ms wall with the triangle upload", flush=True)
    if args.check:
        h = Scene("")
        h.add_triangles(tris)
        h.build(args.threads)
        _, ht, hn = h.arrays()
        assert gn.tobytes() == hn.tobytes() and gt.tobytes() == ht.tobytes(), "GPU and host builds differ"
        print("  identical to the host build of the same input")
    wf.close()


if __name__ == "__main__":
    main()
