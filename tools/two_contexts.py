#!/usr/bin/env python3
"""Experiment: do two render pipelines on one GPU overlap usefully?  Two contexts render the bundled frame concurrently (their
kernels meet on the device: one context's shade beside the other's extend) against one context rendering twice as many frames.

  python tools/two_contexts.py [--blocks 6] [--frames 6] [--spp 16]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=0)
    ap.add_argument("--frames", type=int, default=6)
    ap.add_argument("--spp", type=int, default=16)
    ap.add_argument("--contexts", type=int, default=2)
    args = ap.parse_args()
    import pathtracer_amd as pa
    from pathtracer_amd.host import Scene, Camera
    s = Scene()
    s.initialize_scene(11)
    s.build(threads=8)
    mats, tris, nodes = s.arrays()
    w, h = 1920, 1080
    ctxs = []
    for k in range(args.contexts):
        wf = pa.Wavefront(nee=True, rng_mode=1, stage_timing=0)
        wf.upload_scene(mats, tris, nodes)
        if args.blocks:
            wf.set_option(6, args.blocks)
        wf.allocate(w, h)
        ctxs.append(wf)
    fs = Camera(w, h, None).frame_state((0, 2, 15), (0, 3, 0), frames=1, samples=0)

    def run(active, frames, spp):
        for wf in active:
            wf.invoke(fs, spp=spp, max_bounces=4, wait=False)
        for wf in active:
            wf.synchronize()
        t0 = time.perf_counter()
        for _ in range(frames):
            for wf in active:
                wf.invoke(fs, spp=spp, max_bounces=4, wait=False)
        for wf in active:
            wf.synchronize()
        return (time.perf_counter() - t0) * 1e3

    one = run(ctxs[:1], args.frames, args.spp)
    print(f"one context, {args.frames} frames of {args.spp} spp: {one / args.frames:.3f} ms per frame", flush=True)
    half = args.spp // args.contexts
    both = run(ctxs, args.frames, half)
    print(f"{args.contexts} contexts, {args.frames} frames of {half} spp each, concurrently: {both / args.frames:.3f} ms per {args.spp}-spp frame equivalent "
          f"({one / both:.3f}x)", flush=True)
    # the contexts render whole frames side by side: what pipelining consecutive frames of one context would give
    full = run(ctxs, args.frames, args.spp)
    print(f"{args.contexts} contexts, {args.frames} frames of {args.spp} spp each, concurrently: {full / args.frames / args.contexts:.3f} ms per frame "
          f"({one * args.contexts / full:.3f}x)", flush=True)
    for wf in ctxs:
        wf.close()


if __name__ == "__main__":
    main()
