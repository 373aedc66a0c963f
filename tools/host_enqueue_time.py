#!/usr/bin/env python3
"""How long the host takes to queue a frame against how long the GPU takes to render it (small frames are bound by the former).
   python tools/host_enqueue_time.py"""
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
    for (w, h, spp, bounces, nee) in [(512, 512, 1, 2, False), (1920, 1080, 1, 4, True), (1920, 1080, 16, 4, True)]:
        wf = pa.Wavefront(nee=nee, rng_mode=1)
        wf.upload_scene(mats, tris, nodes)
        wf.allocate(w, h)
        fs = Camera(w, h, None).frame_state((0, 2, 15), (0, 3, 0), frames=1, samples=0)
        for _ in range(4):
            wf.invoke(fs, spp=spp, max_bounces=bounces, wait=False)
        wf.synchronize()
        n = 200 if w < 1000 else 40
        t0 = time.perf_counter()
        for _ in range(n):
            wf.invoke(fs, spp=spp, max_bounces=bounces, wait=False)
        t1 = time.perf_counter()
        wf.synchronize()
        t2 = time.perf_counter()
        print(f"{w}x{h} x {spp} spp, {bounces} bounces: host queues a frame in {(t1 - t0) * 1e3 / n:.3f} ms, frames complete every {(t2 - t0) * 1e3 / n:.3f} ms", flush=True)
        wf.close()


if __name__ == "__main__":
    main()
