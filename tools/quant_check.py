#!/usr/bin/env python3
"""Quantised (64-byte) against exact (128-byte) node records on the GPU: closest hits, occlusion flags and a small image must be
the same bits whichever records the kernels walk; prints both extend rates on incoherent rays."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtracer_amd as pa
from pathtracer_amd.host import Scene, Camera


def rays_in(lo, hi, n, seed):
    rng = np.random.RandomState(seed)
    o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    return np.concatenate([o, d], 1).astype(np.float32)


def main():
    ok = True
    for name in sys.argv[1:] or ["bundled", "soup200k", "soup1m"]:
        if name == "bundled":
            s = Scene(); s.initialize_scene(11)
            lo, hi, view = np.array([-8, 0, -8], np.float32), np.array([8, 8, 8], np.float32), ((0, 2, 15), (0, 3, 0))
        else:
            s = Scene(""); s.generate_soup({"soup200k": 200000, "soup1m": 1000000}[name])
            lo, hi, view = np.array([-50] * 3, np.float32), np.array([50] * 3, np.float32), ((0, 0, 120), (0, 0, 0))
        s.build(threads=8)
        mats, tris, nodes = s.arrays()
        wf = pa.Wavefront(nee=True, rng_mode=1)
        wf.upload_scene(mats, tris, nodes)
        n = 2000000
        rays = rays_in(lo, hi, n, 3)
        far = rays_in(lo * 6, hi * 6, 200000, 4)        # origins outside the scene, some beyond the grid's reach
        tmax = np.random.RandomState(5).uniform(1, 60, n).astype(np.float32)
        res = {}
        for mode in (0, 1):
            wf.set_option(9, mode)
            h = wf.trace_closest(rays)
            hf = wf.trace_closest(far)
            occ = wf.trace_occluded(rays, tmax)
            wf.allocate(256, 144)
            fs = Camera(256, 144).frame_state(view[0], view[1], frames=1)
            wf.reset_stats()
            wf.invoke(fs, spp=2, max_bounces=4)
            img = wf.read_image()
            st = wf.stats()
            t0 = time.time()
            ms, _ = wf.bench_extend(rays, iters=3)
            res[mode] = (h, hf, occ, img, st, ms)
        a, b = res[0], res[1]
        same = [np.array_equal(a[0][k], b[0][k]) for k in ("tri", "t")] + [np.array_equal(a[1][k], b[1][k]) for k in ("tri", "t")]
        same += [np.array_equal(a[2], b[2]), np.array_equal(a[3].view(np.uint32), b[3].view(np.uint32))]
        same += [a[4]["extend_rays"] == b[4]["extend_rays"], a[4]["shadow_rays"] == b[4]["shadow_rays"], b[4]["stack_overflows"] == 0]
        diff = int((a[0]["tri"] != b[0]["tri"]).sum())
        print(f"{name}: closest tri/t, far tri/t, occluded, image, extend rays, shadow rays, no overflow: {same}; differing ids {diff} of {n}; "
              f"special rays exact {a[4]['special_rays']} quantised {b[4]['special_rays']}; "
              f"extend {n / a[5] / 1e3:.0f} Mrays/s exact, {n / b[5] / 1e3:.0f} Mrays/s quantised", flush=True)
        ok = ok and all(same)
        wf.close()
    print("OK" if ok else "MISMATCH")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
