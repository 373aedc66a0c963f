#!/usr/bin/env python3
"""Frames of the default bench workload rendered two ways: each frame waited for before the next is queued (a host that displays
every frame), and all frames queued back to back (an offline accumulation).   python tools/frame_sync_bench.py [--no-pipeline]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import pathtracer_amd as pa
    from pathtracer_amd.host import Scene, Camera
    s = Scene()
    s.initialize_scene(11)
    s.build(threads=8)
    mats, tris, nodes = s.arrays()
    for (w, h, spp, bounces, nee) in [(1920, 1080, 16, 4, True), (512, 512, 1, 2, False)]:
        wf = pa.Wavefront(nee=nee, rng_mode=1)
        wf.upload_scene(mats, tris, nodes)
        if "--no-pipeline" in sys.argv:
            wf.set_option(11, 0)
        wf.allocate(w, h)
        fs = Camera(w, h, None).frame_state((0, 2, 15), (0, 3, 0), frames=1, samples=0)
        n = 12
        for mode in ("waited", "queued"):
            for _ in range(2):
                wf.invoke(fs, spp=spp, max_bounces=bounces, wait=True)
            t0 = time.perf_counter()
            for _ in range(n):
                wf.invoke(fs, spp=spp, max_bounces=bounces, wait=(mode == "waited"))
            wf.synchronize()
            print(f"{w}x{h} x {spp} spp, frames {mode}: {(time.perf_counter() - t0) * 1e3 / n:.3f} ms per frame", flush=True)
        wf.close()


if __name__ == "__main__":
    main()
