#!/usr/bin/env python3
"""Writes the committed fixtures.

  pathtracer_amd/assets/meshes.bin   parsed arrays of the reference's OBJ/MTL assets (needs /root/reference)
  tests/golden/oracle_golden.json    digests of the oracle's bundled scene and primary hits
  tests/golden/oracle_golden.npz     a 64x64 crop of the 512x512 hit-id image and a 64x64 radiance sum
  tests/golden/oracle_golden_mis.npz a 32x32 radiance sum of the Cornell box with NEE + multiple importance sampling

The reference has no golden vectors of its own; these freeze the CPU oracle's outputs (which were
checked against the .NET System.Random known answers, brute-force traversal and BVH invariants in
tests/test_oracle.py) so that later edits cannot move them unnoticed.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import binding as ob  # noqa: E402

REF_ASSETS = "/root/reference/PathTracer/Assets/Mesh"
MESHES = os.path.join(ROOT, "pathtracer_amd", "assets", "meshes.bin")


def main():
    if os.path.isdir(REF_ASSETS):
        ob.write_mesh_blob(REF_ASSETS, ["light", "cube", "sphere", "bunny", "torus", "cornell", "floor"], MESHES)
    s = ob.Scene(MESHES)
    s.initialize_scene(11)
    s.build()
    mats, tris, nodes = s.arrays()
    w = h = 512
    fs, _ = s.camera((0, 2, 15), (0, 3, 0), w, h, use_scene=False, frames=1, samples=1)
    rays, _ = ob.generate(ob.render_state(w, h), fs, ob.path_config(), w * h)
    lit = s.intersect(rays, "literal", threads=8)
    g = {"bundled": {
        "triangles": int(tris.size), "nodes": int(nodes.size), "materials": int(mats.size),
        "materials_sha256": hashlib.sha256(mats.tobytes()).hexdigest(),
        "triangles_sha256": hashlib.sha256(tris.tobytes()).hexdigest(),
        "nodes_sha256": hashlib.sha256(nodes.tobytes()).hexdigest(),
        "primary_512_hit_ids_sha256": hashlib.sha256(lit["tri"].tobytes()).hexdigest(),
        "primary_512_hit_t_sha256": hashlib.sha256(lit["t"].tobytes()).hexdigest(),
        "primary_512_hit_fraction": float((lit["tri"] >= 0).mean()),
    }}
    with open(os.path.join(HERE, "oracle_golden.json"), "w") as f:
        json.dump(g, f, indent=1)
    fs64, _ = s.camera((0, 2, 15), (0, 3, 0), 64, 64, use_scene=False, frames=1, samples=1)
    img, _ = s.render(ob.render_state(64, 64), fs64, ob.path_config(4, 1, 0), 4, threads=8)
    np.savez_compressed(os.path.join(HERE, "oracle_golden.npz"), hit_ids_crop=lit["tri"].reshape(h, w)[224:288, 224:288], radiance_64_sum=img)
    # Cornell box with next-event estimation and multiple importance sampling (the estimator that is not in the reference)
    c = ob.Scene(MESHES)
    c.add_mesh("cornell")
    c.build()
    fsc, _ = c.camera((0, 1, 3.4), (0, 1, 0), 32, 32, use_scene=True, frames=1, samples=1)
    imgc, _ = c.render(ob.render_state(32, 32, sky=(0, 0, 0, 1)), fsc, ob.path_config(5, 2, 1), 8, threads=8)
    np.savez_compressed(os.path.join(HERE, "oracle_golden_mis.npz"), cornell_mis_32_sum=imgc)
    print(json.dumps(g, indent=1))


if __name__ == "__main__":
    main()
