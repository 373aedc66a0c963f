#!/usr/bin/env python3
"""The reference's box test is not conservative with respect to its triangle test: rays that hit a triangle can miss the padded
box of the triangle's leaf, and Accelerator.Intersect (Accelerator.cs:51-111) then never tests the triangle.

4,000 axis-aligned thin triangles at coordinates 60..70 (primitive boxes are padded by 1e-5, Box.cs:95-102; one float ulp at 64 is
7.6e-6), 400,000 rays of length 180 aimed next to triangle vertices: the oracle's brute-force closest hit (every triangle tested)
against its literal BVH walk.  About 1 % of the rays lose their hit in the walk; with rays of length 20 none does.  This is why a
traversal over other boxes (an instanced scene's, a wider tree's) cannot promise the flattened scene's results bit for bit.

  python tests/experiments/leaf_box_gap.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import binding as ob   # noqa: E402
from pathtracer_amd.types import TRIANGLE, material   # noqa: E402


def main():
    rng = np.random.default_rng(1)
    n, m = 4000, 400000
    tris = np.zeros(n, TRIANGLE)
    base = rng.uniform(60, 70, (n, 3)).astype(np.float32)
    flat = rng.integers(0, 3, n)
    for name in ("v1", "v2", "v3"):
        off = rng.uniform(-0.3, 0.3, (n, 3)).astype(np.float32)
        off[np.arange(n), flat] = 0
        tris[name]["position"] = base + off
        tris[name]["normal"] = (0, 1, 0)
    o = ob.Scene("")
    o.set_materials(material())
    o.add_triangles(tris)
    o.build()
    for length, alpha in ((20.0, 1.0), (180.0, 0.02)):
        pick = rng.integers(0, n, m)
        w = rng.dirichlet((alpha,) * 3, m).astype(np.float32)
        tgt = tris["v1"]["position"][pick] * w[:, 0:1] + tris["v2"]["position"][pick] * w[:, 1:2] + tris["v3"]["position"][pick] * w[:, 2:3]
        d = rng.normal(size=(m, 3)).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
        rays = np.concatenate([(tgt - d * np.float32(length)).astype(np.float32), d], 1).astype(np.float32)
        brute, walk = o.intersect(rays, "brute", threads=8), o.intersect(rays, "literal", threads=8)
        lost = int(((brute["tri"] >= 0) & ((walk["tri"] != brute["tri"]) | (walk["t"].view(np.uint32) != brute["t"].view(np.uint32)))).sum())
        print(f"ray length {length:5.0f}, targets {'next to vertices' if alpha < 1 else 'anywhere on the triangle'}: {lost} of {m} rays "
              f"lose or change their hit in the BVH walk ({(brute['tri'] >= 0).mean():.2f} hit something)")


if __name__ == "__main__":
    main()
