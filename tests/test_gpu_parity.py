"""GPU parity: the HIP stages, called through the C ABI, against the CPU oracle on the same inputs.

Bars: hit triangle ids, hit distances, barycentrics, generated rays and occlusion flags are
compared bit for bit.  Radiance is compared with relative RMSE <= 1e-6 AND (stronger) the count
of differing pixels is reported and required to be zero where every operation is an IEEE
+,-,*,/,sqrt (which is the whole pipeline: the transcendentals are fixed polynomials on both sides).
"""
import os

import numpy as np
import pytest

from conftest import bundled_view, rel_rmse

pytestmark = pytest.mark.gpu

RADIANCE_RELRMSE_TOL = 1e-6


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def pa():
    import pathtracer_amd
    return pathtracer_amd


def make_wf(pa, scene, **kw):
    wf = pa.Wavefront(**kw)
    mats, tris, nodes = scene.arrays()
    wf.upload_scene(mats, tris, nodes)
    return wf


@pytest.fixture(scope="module")
def wf_bundled(pa, bundled):
    wf = make_wf(pa, bundled)
    yield wf
    wf.close()


def test_generate_bit_exact(pa, ob, bundled, wf_bundled):
    for (w, h, frames) in [(512, 512, 1), (512, 512, 0), (64, 40, 7), (60, 36, 3)]:
        wf_bundled.allocate(w, h)
        fs = bundled_view(bundled, w, h, frames=frames)
        n = ((w + 7) // 8) * ((h + 7) // 8) * 64
        for mode in (0, 1):
            wf_bundled.set_option(2, mode)
            rays, pix = wf_bundled.generate_rays(fs, n)
            orays, opix = ob.generate(ob.render_state(w, h), fs, ob.path_config(rng_mode=mode), n)
            assert np.array_equal(pix, opix)
            assert np.array_equal(bits(rays), bits(orays)), f"generate differs at {w}x{h} frames={frames} mode={mode}"
    wf_bundled.set_option(2, 0)


def test_primary_hit_ids_bit_exact_bundled_512(ob, bundled, wf_bundled):
    """The reference's own setting: bundled scene, 512x512 primary rays; ids/t vs the literal C# traversal."""
    w = h = 512
    fs = bundled_view(bundled, w, h, frames=1)
    rays, pix = ob.generate(ob.render_state(w, h), fs, ob.path_config(), w * h)
    g = wf_bundled.trace_closest(rays)
    lit = bundled.intersect(rays, "literal", threads=8)
    can = bundled.intersect(rays, "canonical", threads=8)
    assert (lit["tri"] >= 0).mean() > 0.3
    assert np.array_equal(g["tri"], can["tri"])
    assert np.array_equal(bits(g["t"]), bits(can["t"]))
    assert np.array_equal(bits(g["uv"]), bits(can["uv"]))
    # the contract: identical to the reference's own CPU traversal
    assert np.array_equal(g["tri"], lit["tri"])
    assert np.array_equal(bits(g["t"]), bits(lit["t"]))


def test_incoherent_rays_soup(ob, soup50k, pa):
    wf = make_wf(pa, soup50k)
    rng = np.random.RandomState(5)
    n = 200000
    o = rng.uniform(-60, 60, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([o, d.astype(np.float32)], 1)
    g = wf.trace_closest(rays)
    can = soup50k.intersect(rays, "canonical", threads=8)
    lit = soup50k.intersect(rays, "literal", threads=8)
    assert (can["tri"] >= 0).mean() > 0.1
    assert np.array_equal(g["tri"], can["tri"]) and np.array_equal(bits(g["t"]), bits(can["t"]))
    assert np.array_equal(g["tri"], lit["tri"]) and np.array_equal(bits(g["t"]), bits(lit["t"]))
    tmax = rng.uniform(1, 80, n).astype(np.float32)
    assert np.array_equal(wf.trace_occluded(rays, tmax), soup50k.occluded(rays, tmax, threads=8))
    assert wf.stats()["stack_overflows"] == 0
    wf.close()


def test_axis_aligned_and_degenerate_rays(ob, bundled, wf_bundled):
    """Zero direction components make 0*inf = NaN slab distances; MathF.Min/Max propagate them."""
    rays = []
    # the autofocus ray of Camera.GetFocalDistance: forward of (0,2,15)->(0,3,0) has x == 0
    f = np.array([0, 1, -15], np.float32)
    f = f / np.sqrt(np.float32(f @ f))
    rays.append([0, 2, 15, *f])
    for x in range(-3, 4):
        for z in range(-3, 4):
            rays.append([x, 5, z, 0, -1, 0])          # straight down onto tile corners / edges
            rays.append([x + 0.5, 5, z + 0.25, 0, -1, 0])
            rays.append([x, 0, z, 1, 0, 0])           # in the floor plane
            rays.append([x, 1, z, 0, 0, 0])           # null direction
    rays.append([np.nan, 0, 0, 0, -1, 0])
    rays.append([0, 5, 0, np.inf, -1, 0])
    rays = np.array(rays, np.float32)
    g = wf_bundled.trace_closest(rays)
    can = bundled.intersect(rays, "canonical")
    lit = bundled.intersect(rays, "literal")
    assert np.array_equal(g["tri"], can["tri"]) and np.array_equal(bits(g["t"]), bits(can["t"]))
    # Rays aimed exactly at shared vertices / edges hit several triangles at bit-identical
    # distances.  The C# traversal keeps the first one it happens to test; the contract here is
    # the smallest triangle index.  Distances must agree everywhere, ids wherever there is no tie.
    assert np.array_equal(bits(g["t"]), bits(lit["t"]))
    differs = np.nonzero(g["tri"] != lit["tri"])[0]
    assert len(differs) > 0, "the corner rays are meant to produce exact ties"
    brute = bundled.intersect(rays[differs], "brute")
    assert np.array_equal(brute["tri"], g["tri"][differs])
    assert (g["tri"][differs] < lit["tri"][differs]).all()


def test_special_rays_in_the_extend_stage(ob, bundled, wf_bundled):
    """Rays with NaN-capable slab distances take the two-wide step inside the wide kernel: a queue that mixes both kinds
    (both kinds in one wave) must come out exactly as the per-ray queries do."""
    rng = np.random.default_rng(5)
    n = 4096
    rays = np.zeros((n, 6), np.float32)
    rays[:, 0:3] = rng.uniform(-8, 8, (n, 3)).astype(np.float32) + np.array([0, 9, 0], np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    rays[:, 3:6] = d / np.sqrt((d * d).sum(1, dtype=np.float32))[:, None]
    special = rng.random(n) < 0.3
    axis = rng.integers(0, 3, n)
    rays[special, 3 + axis[special]] = 0.0          # zero direction component
    rays[rng.random(n) < 0.02, 3:6] = [0, -1, 0]    # straight down
    _, hits = wf_bundled.bench_extend(rays, iters=1)
    tri, t = hits["tri"], hits["t"]
    can = bundled.intersect(rays, "canonical")
    lit = bundled.intersect(rays, "literal")
    assert np.array_equal(tri, can["tri"]) and np.array_equal(bits(t), bits(can["t"]))
    assert np.array_equal(bits(t), bits(lit["t"]))
    assert special.sum() > 1000 and (tri >= 0).sum() > 1000


def test_quantised_node_records(pa, ob, bundled, soup50k):
    """The 64-byte node records (boxes on a 16-bit grid, exact leaf boxes applied at the triangles) are what the kernels walk for
    trees beyond L2; forced on here for two small scenes.  Hits, distances, barycentrics, occlusion flags, a frame with next-event
    estimation and the step / triangle-test counters of both traversal stages equal the oracle's restatement of that walk — and the
    hits equal the literal C# traversal, which knows nothing of the grid.  Rays that start far outside the scene (beyond the grid's
    reach), with tiny direction components or NaN-capable ones fall back to the exact two-wide walk in the same waves."""
    for scene, lo, hi, view in [(bundled, (-8, 0, -8), (8, 8, 8), ((0, 2, 15), (0, 3, 0))), (soup50k, (-60, -60, -60), (60, 60, 60), ((0, 0, 120), (0, 0, 0)))]:
        wf = make_wf(pa, scene, nee=True, rng_mode=1, traversal_stats=True)
        wf.set_option(9, 1)
        scene.set_quantized(1)
        try:
            rng = np.random.default_rng(11)
            n = 120000
            rays = np.zeros((n, 6), np.float32)
            rays[:, 0:3] = rng.uniform(lo, hi, (n, 3))
            d = rng.normal(size=(n, 3)).astype(np.float32)
            rays[:, 3:6] = d / np.sqrt((d * d).sum(1, dtype=np.float32))[:, None]
            far = rng.random(n) < 0.1                       # origins up to 60 scene extents away: beyond the grid's reach of 16
            rays[far, 0:3] *= rng.uniform(1, 400, (int(far.sum()), 1)).astype(np.float32)
            aim = -rays[far, 0:3] + rng.uniform(lo, hi, (int(far.sum()), 3)).astype(np.float32)
            rays[far, 3:6] = aim / np.sqrt((aim * aim).sum(1, dtype=np.float32))[:, None]
            tiny = rng.random(n) < 0.05
            rays[tiny, 3 + rng.integers(0, 3, int(tiny.sum()))] = np.float32(1e-25)   # below the quantised walk's 1e-20
            zero = rng.random(n) < 0.05
            rays[zero, 3 + rng.integers(0, 3, int(zero.sum()))] = 0.0
            wf.reset_stats()
            g = wf.trace_closest(rays)
            assert wf.stats()["special_rays"] > 0.08 * n
            can = scene.intersect(rays, "canonical", threads=8)
            lit = scene.intersect(rays, "literal", threads=8)
            assert (g["tri"] >= 0).mean() > 0.1
            assert np.array_equal(g["tri"], can["tri"]) and np.array_equal(bits(g["t"]), bits(can["t"])) and np.array_equal(bits(g["uv"]), bits(can["uv"]))
            assert np.array_equal(bits(g["t"]), bits(lit["t"]))
            differs = np.nonzero(g["tri"] != lit["tri"])[0]      # exact distance ties only, towards the smaller index
            assert (g["tri"][differs] < lit["tri"][differs]).all() and len(differs) <= 8
            tmax = rng.uniform(1, 80, n).astype(np.float32)
            assert np.array_equal(wf.trace_occluded(rays, tmax), scene.occluded(rays, tmax, threads=8))
            w, h, spp = 192, 128, 2
            wf.allocate(w, h)
            fs, _ = scene.camera(view[0], view[1], w, h, use_scene=False, frames=1, samples=1)
            wf.reset_stats()
            wf.invoke(fs, spp=spp, max_bounces=4)
            img = wf.read_image()
            st = wf.stats()
            osum, ost = scene.render(ob.render_state(w, h), fs, ob.path_config(4, 1, 1), spp, threads=8)
            assert st["stack_overflows"] == 0
            for k in ["extend_rays", "shadow_rays", "extend_steps", "extend_tris", "shadow_steps", "shadow_tris"]:
                assert st[k] == ost[k], (k, st[k], ost[k])
            assert np.array_equal(bits(img), bits(osum / np.float32(spp)))
            # and the image is the one the exact records give
            wf.set_option(9, 0)
            wf.allocate(w, h)
            wf.invoke(fs, spp=spp, max_bounces=4)
            assert np.array_equal(bits(wf.read_image()), bits(img))
        finally:
            scene.set_quantized(2)
            wf.close()


def _host_build(pa, tris, threads=4):
    from pathtracer_amd.host import Scene
    s = Scene("")
    s.add_triangles(tris)
    s.build(threads)
    _, t, n = s.arrays()
    return t, n


def test_gpu_bvh_build_is_byte_identical(pa, bundled, soup50k):
    """pt_build_bvh against the oracle's literal Accelerator.Build (insertion-order references in, same Node[] and
    Triangle[] bytes out) and, on further inputs, against the host mirror of the same build."""
    wf = pa.Wavefront()
    rng = np.random.default_rng(3)
    for scene in (bundled, soup50k):
        _, want_t, want_n, refs = scene.arrays(references=True)
        got_t, got_n, _ = wf.build_bvh(refs)
        assert got_n.tobytes() == want_n.tobytes() and got_t.tobytes() == want_t.tobytes()
    bt = bundled.arrays()[1]
    cases = {"bundled shuffled": bt[rng.permutation(bt.size)], "two": bt[:2], "three": bt[:3], "seven": bt[100:107]}
    # degenerate inputs: identical triangles (no centroid spread: Accelerator.cs:203-204), and a row of equal boxes
    same = np.repeat(bt[5:6], 40)
    cases["identical"] = same
    row = np.repeat(bt[5:6], 64)
    for k in range(64):
        for v in ("v1", "v2", "v3"):
            row[v]["position"][k, 0] += np.float32(k // 8)       # eight clusters of eight identical triangles
    cases["clusters"] = row
    for name, tris in cases.items():
        want_t, want_n = _host_build(pa, tris)
        got_t, got_n, ms = wf.build_bvh(tris)
        assert got_n.size == want_n.size, name
        assert got_n.tobytes() == want_n.tobytes(), name
        assert got_t.tobytes() == want_t.tobytes(), name
    # large nodes (the chunked partition) whose splits fail: 3,000 copies each of two triangles -> one split, two huge leaves
    two = np.concatenate([np.repeat(bt[5:6], 3000), np.repeat(bt[9000:9001], 3000)])
    two = two[rng.permutation(two.size)]
    want_t, want_n = _host_build(pa, two)
    got_t, got_n, _ = wf.build_bvh(two)
    assert got_n.tobytes() == want_n.tobytes() and got_t.tobytes() == want_t.tobytes()
    assert (want_n["count"] > 15).any(), "meant to produce leaves too large for a packed entry"
    # ... and such leaves traverse correctly (leaf phases split a 3,000-triangle leaf over many rounds)
    from pathtracer_amd.host import Scene
    hs = Scene("")
    hs.add_triangles(two)
    hs.build(1)
    big = pa.Wavefront()
    big.upload_scene(bundled.arrays()[0], *hs.arrays()[1:])
    c = (two["v1"]["position"] + two["v2"]["position"] + two["v3"]["position"]) / np.float32(3)
    o = c[:256] + np.array([0.3, 4.0, 0.2], np.float32)
    d = c[:256] - o
    rays = np.concatenate([o, d / np.sqrt((d * d).sum(1, dtype=np.float32))[:, None]], axis=1).astype(np.float32)
    g = big.trace_closest(rays)
    assert (g["tri"] >= 0).all()
    # every copy is hit at the same distance: the smallest index among the copies wins
    first_of = {}
    for i, t in enumerate(want_t):
        first_of.setdefault(t["v1"]["position"].tobytes(), i)
    assert set(g["tri"].tolist()) <= set(first_of.values())
    big.close()
    # non-finite coordinates (the C# build throws IndexOutOfRange): no fault here, every triangle still lands in a leaf
    bad = bt[:5000].copy()
    bad["v2"]["position"][7, 1] = np.nan
    bad["v1"]["position"][99, 0] = np.inf
    _, bn, _ = wf.build_bvh(bad)
    assert bn[bn["count"] > 0]["count"].sum() == bad.size
    t0, n0, _ = wf.build_bvh(bt[:0])
    assert t0.size == 0 and n0.size == 1 and n0["count"][0] == 0
    wf.close()


def test_device_built_light_table_and_small_subtrees(pa, ob):
    """pt_build_scene on little scenes that sit on the seams of the device code: triangle counts around the small-subtree
    limit (one wave finishes a node of <= 64 primitives) and emissive-triangle counts around the 64-lane rounds of the running
    area sum.  Arrays == the host build's; a next-event-estimation frame == the same scene uploaded from the host (whose light
    table is summed on the CPU, one float addition after the other)."""
    from pathtracer_amd.types import TRIANGLE, material, DIFFUSE
    rng = np.random.default_rng(21)
    mats = np.concatenate([material(color=(0.7, 0.7, 0.7, 0)), material(color=(1, 1, 1, 0), emissive=(4, 3, 2, 0))])
    for n, n_lights in ((3, 1), (63, 63), (64, 64), (65, 65), (129, 128), (700, 193), (5000, 1500)):
        tris = np.zeros(n, TRIANGLE)
        c = rng.uniform(-4, 4, (n, 3)).astype(np.float32)
        for v in ("v1", "v2", "v3"):
            tris[v]["position"] = c + rng.uniform(-0.4, 0.4, (n, 3)).astype(np.float32)
            tris[v]["normal"] = (0, 1, 0)
        tris["material"] = 0
        tris["material"][rng.permutation(n)[:n_lights]] = 1
        want_t, want_n = _host_build(pa, tris)
        dev = pa.Wavefront(nee=True, rng_mode=1)
        got_t, got_n, _ = dev.build_scene(mats, tris, want_arrays=True)
        assert got_n.tobytes() == want_n.tobytes() and got_t.tobytes() == want_t.tobytes(), n
        ref = pa.Wavefront(nee=True, rng_mode=1)
        ref.upload_scene(mats, want_t, want_n)
        o = ob.Scene("")
        fs, _ = o.camera((0, 0, 14), (0, 0, 0), 64, 48, use_scene=False, frames=1, samples=1)
        imgs = []
        for x in (dev, ref):
            x.allocate(64, 48)
            x.reset_stats()
            x.invoke(fs, spp=4, max_bounces=3)
            imgs.append(x.read_image())
            assert n < 700 or x.stats()["shadow_rays"] > 100
            x.close()
        assert np.array_equal(bits(imgs[0]), bits(imgs[1])), (n, n_lights)


def test_device_built_scene_renders_identically(pa, ob, bundled, wf_bundled):
    """pt_build_scene (build + repack on the GPU) against pt_upload_scene of the CPU-built arrays: same hits, same
    counters, same image, and the arrays handed back are the CPU build's."""
    mats, tris, nodes, refs = bundled.arrays(references=True)
    wf = pa.Wavefront(nee=True, rng_mode=1, traversal_stats=True)
    got_t, got_n, ms = wf.build_scene(mats, refs, want_arrays=True)
    assert got_t.tobytes() == tris.tobytes() and got_n.tobytes() == nodes.tobytes()
    rng = np.random.default_rng(11)
    n = 20000
    rays = np.zeros((n, 6), np.float32)
    rays[:, 0:3] = rng.uniform(-8, 8, (n, 3)).astype(np.float32) + np.array([0, 9, 0], np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    rays[:, 3:6] = d / np.sqrt((d * d).sum(1, dtype=np.float32))[:, None]
    rays[::7, 4] = 0.0                      # some rays for the two-wide walk (pair records)
    a, b = wf.trace_closest(rays), wf_bundled.trace_closest(rays)
    for k in ("tri", "t", "uv"):
        assert np.array_equal(bits(a[k]), bits(b[k])), k
    tmax = rng.uniform(1, 30, n).astype(np.float32)
    assert np.array_equal(wf.trace_occluded(rays, tmax), wf_bundled.trace_occluded(rays, tmax))
    w = h = 128
    ref = pa.Wavefront(nee=True, rng_mode=1, traversal_stats=True)
    ref.upload_scene(mats, tris, nodes)
    fs = bundled_view(bundled, w, h)
    imgs, stats = [], []
    for x in (wf, ref):
        x.allocate(w, h)
        x.reset_stats()
        x.invoke(fs, spp=2, max_bounces=4)
        imgs.append(x.read_image())
        stats.append(x.stats())
    assert np.array_equal(bits(imgs[0]), bits(imgs[1]))
    for k in ("extend_rays", "shadow_rays", "extend_steps", "extend_tris", "shadow_steps", "shadow_tris"):
        assert stats[0][k] == stats[1][k], k
    # an empty and a two-triangle scene through the same entry point
    wf.build_scene(mats, refs[:0])
    assert (wf.trace_closest(rays[:8])["tri"] == -1).all()
    wf.build_scene(mats, refs[:2])
    wf.close()
    ref.close()


def test_against_committed_golden_fixtures(pa):
    """The product alone (host mirror + HIP path, no oracle in the process of this test) against tests/golden/: scene
    digests, the 512x512 primary hit image of the literal C# traversal, and two frozen radiance sums (bundled scene with
    NEE; Cornell box with multiple importance sampling)."""
    import hashlib
    import json
    from pathtracer_amd.host import Scene, Camera
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = json.load(open(os.path.join(golden, "oracle_golden.json")))["bundled"]
    z = np.load(os.path.join(golden, "oracle_golden.npz"))
    zm = np.load(os.path.join(golden, "oracle_golden_mis.npz"))
    s = Scene()
    s.initialize_scene(11)
    s.build()
    mats, tris, nodes = s.arrays()
    assert hashlib.sha256(mats.tobytes()).hexdigest() == g["materials_sha256"]
    assert hashlib.sha256(tris.tobytes()).hexdigest() == g["triangles_sha256"]
    assert hashlib.sha256(nodes.tobytes()).hexdigest() == g["nodes_sha256"]
    wf = pa.Wavefront(nee=True, rng_mode=0)
    wf.upload_scene(mats, tris, nodes)
    w = h = 512
    wf.allocate(w, h)
    fs = Camera(w, h).frame_state((0, 2, 15), (0, 3, 0), frames=1, samples=1)
    rays, _ = wf.generate_rays(fs, w * h)
    hit = wf.trace_closest(rays)
    assert hashlib.sha256(hit["tri"].tobytes()).hexdigest() == g["primary_512_hit_ids_sha256"]
    assert hashlib.sha256(hit["t"].tobytes()).hexdigest() == g["primary_512_hit_t_sha256"]
    assert np.array_equal(hit["tri"].reshape(h, w)[224:288, 224:288], z["hit_ids_crop"])
    wf.allocate(64, 64)
    wf.invoke(Camera(64, 64).frame_state((0, 2, 15), (0, 3, 0), frames=1, samples=1), spp=4, max_bounces=4)
    assert np.array_equal(bits(wf.read_image() * np.float32(4)), bits(z["radiance_64_sum"]))
    wf.close()

    c = Scene()
    c.add_mesh("cornell")
    c.build()
    wf = pa.Wavefront(nee=2, rng_mode=1)
    wf.upload_scene(*c.arrays())
    wf.allocate(32, 32, sky=(0, 0, 0, 1))
    fs = Camera(32, 32, wf).frame_state((0, 1, 3.4), (0, 1, 0), frames=1, samples=1)
    wf.invoke(fs, spp=8, max_bounces=5)
    assert np.array_equal(bits(wf.read_image() * np.float32(8)), bits(zm["cornell_mis_32_sum"]))
    wf.close()


def test_tiny_and_empty_scenes(pa, ob):
    # 2 triangles: the root stays a leaf (Count < 3, Accelerator.cs:138)
    s = ob.Scene(pa_assets())
    s.add_mesh("light")
    s.build()
    assert s.counts()["nodes"] == 1
    wf = make_wf(pa, s)
    rays = np.array([[0, 0, 0, 0, 1, 0], [0, 0, 0, 0, -1, 0], [4.9, 0, 4.9, 0.001, 1, 0.001], [20, 0, 0, 0, 1, 0]], np.float32)
    g = wf.trace_closest(rays)
    lit = s.intersect(rays, "literal")
    assert np.array_equal(g["tri"], lit["tri"]) and np.array_equal(bits(g["t"]), bits(lit["t"]))
    assert g["tri"][0] >= 0 and g["tri"][1] == -1
    # empty scene: everything misses, a render is all sky
    wf.upload_scene(np.zeros(0, pa.MATERIAL), np.zeros(0, pa.TRIANGLE), np.zeros(0, pa.NODE))
    g = wf.trace_closest(rays)
    assert (g["tri"] == -1).all() and np.isinf(g["t"]).all()
    wf.allocate(16, 8)
    fs = bundled_view(s, 16, 8)
    wf.invoke(fs, spp=2, max_bounces=3)
    img = wf.read_image()
    assert np.allclose(img, np.array(pa.SKY_COLOR, np.float32)[None, None, :])
    wf.close()


def test_random_scenes_of_many_sizes(pa, ob):
    """Ragged inputs: random triangle sets of 1 .. 4099 triangles, with zero-area, duplicated and far-away triangles
    mixed in.  For every size the GPU-built tree equals the oracle's literal build byte for byte, and closest / any hits
    of rays aimed through the set (some axis-aligned, some starting on a triangle) equal the oracle's bit for bit."""
    rng = np.random.default_rng(77)
    mats = np.zeros(1, pa.MATERIAL)
    mats["color"] = 0.7
    wf = pa.Wavefront()
    for n in (1, 2, 3, 5, 17, 64, 257, 1000, 4099):
        tris = np.zeros(n, pa.TRIANGLE)
        c = rng.uniform(-4, 4, (n, 1, 3)).astype(np.float32)
        v = c + rng.uniform(-0.8, 0.8, (n, 3, 3)).astype(np.float32)
        if n >= 5:
            v[1] = v[0]                                   # a duplicate
            v[2, 1] = v[2, 0]                             # zero area: two equal vertices
            v[3, :, :] = v[3, 0:1, :]                     # a point
            v[4] += np.float32(300.0)                     # an outlier that stretches the root box
        for k, name in enumerate(("v1", "v2", "v3")):
            tris[name]["position"] = v[:, k, :]
            tris[name]["normal"] = (0, 1, 0)
        s = ob.Scene("")
        s.set_materials(mats)
        s.add_triangles(tris)
        s.build()
        omats, otris, onodes, refs = s.arrays(references=True)
        gt, gn, _ = wf.build_bvh(refs)
        assert gt.tobytes() == otris.tobytes() and gn.tobytes() == onodes.tobytes(), n
        wf.upload_scene(omats, otris, onodes)
        m = 2048
        o = rng.uniform(-6, 6, (m, 3)).astype(np.float32)
        d = (rng.uniform(-4, 4, (m, 3)).astype(np.float32) - o)
        d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
        d[::16] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, len(d[::16]))]          # axis-aligned (zero components)
        o[1::16] = v[rng.integers(0, n, len(o[1::16])), 0, :]                             # starting on a vertex
        rays = np.concatenate([o, d], 1)
        g = wf.trace_closest(rays)
        can = s.intersect(rays, "canonical")
        assert np.array_equal(g["tri"], can["tri"]) and np.array_equal(bits(g["t"]), bits(can["t"])), n
        lit = s.intersect(rays, "literal")
        assert np.array_equal(bits(g["t"]), bits(lit["t"])), n       # ids may differ from the literal order only on exact ties
        tmax = rng.uniform(0.5, 12, m).astype(np.float32)
        assert np.array_equal(wf.trace_occluded(rays, tmax), s.occluded(rays, tmax)), n
    assert wf.stats()["stack_overflows"] == 0
    wf.close()


def test_deep_tree_uses_the_spilled_stack(pa, ob):
    """Triangles on a geometric progression along x give a chain-like tree (depth > 40).  Rays that run along the chain,
    almost parallel to the axis, pass through every box and push more entries than the 10 a lane keeps in LDS, so the
    scratch part of the stack carries real traffic.  Hits and occlusion must still equal the oracle's, with no overflow."""
    n = 120
    x = (2.0 ** np.arange(n)).astype(np.float32)
    tris = np.zeros(n, ob.TRIANGLE)
    for name, off in (("v1", (0, 0, -0.15)), ("v2", (0, 0.3, -0.15)), ("v3", (0, 0, 0.15))):   # across the axis
        p = np.zeros((n, 3), np.float32)
        p[:, 0] = x
        tris[name]["position"] = p + np.float32(off)
        tris[name]["normal"] = (1, 0, 0)
    s = ob.Scene("")
    s.set_materials(np.zeros(1, ob.MATERIAL))
    s.add_triangles(tris)
    s.build()
    assert s.counts()["max_depth"] > 40
    mats, t2, nodes = s.arrays()
    wf = pa.Wavefront(traversal_stats=True)
    wf.upload_scene(mats, t2, nodes)
    rng = np.random.default_rng(9)
    m = 4096
    rays = np.zeros((m, 6), np.float32)
    forward = rng.random(m) < 0.5
    rays[:, 0] = np.where(forward, -rng.uniform(0.5, 100, m), x[rng.integers(10, 100, m)] * 1.5)
    rays[:, 1] = rng.uniform(0.001, 0.12, m)
    rays[:, 2] = rng.uniform(-0.14, 0.02, m)
    rays[:, 3] = np.where(forward, 1.0, -1.0)
    rays[:, 4] = rng.uniform(-1, 1, m) * 1e-30           # tiny but not special: the ray stays inside the chain's boxes
    rays[:, 5] = np.where(rng.random(m) < 0.5, 2e-37, -2e-37)
    g = wf.trace_closest(rays)
    can = s.intersect(rays, "canonical", stats=True)
    assert np.array_equal(g["tri"], can["tri"]) and np.array_equal(bits(g["t"]), bits(can["t"]))
    assert (g["tri"] >= 0).sum() > m // 4
    assert can["stats"]["steps"] > 12 * m                # the rays really walk the chain
    tmax = np.where(forward, rng.uniform(0.5, 300, m), x[rng.integers(10, 100, m)]).astype(np.float32)
    assert np.array_equal(wf.trace_occluded(rays, tmax), s.occluded(rays, tmax))
    st = wf.stats()
    assert st["stack_overflows"] == 0 and st["special_rays"] == 0
    wf.close()


def test_big_leaves(pa, ob):
    """Coincident triangles cannot be split by SAH: leaves of 150 and 17 triangles (beyond the 15 a packed
    stack entry holds, and the larger one beyond the 64 triangles one leaf phase deals) next to ordinary ones; ties among
    the copies go to the smallest index."""
    rng = np.random.RandomState(3)
    big_n = 150
    tris = np.zeros(big_n + 17 + 200, ob.TRIANGLE)

    def put(k, a, b, c):
        n = np.cross(b - a, c - a)
        n = (n / np.linalg.norm(n)).astype(np.float32)
        for name, p in (("v1", a), ("v2", b), ("v3", c)):
            tris[name]["position"][k] = p
            tris[name]["normal"][k] = n
    base = np.array([[0, 0, 0], [2, 0, 0], [0, 2, 0]], np.float32)
    for k in range(big_n):
        put(k, *base)
    for k in range(big_n, big_n + 17):
        put(k, *(base + np.float32([5, 0, 1])))
    for k in range(big_n + 17, tris.size):
        c = rng.uniform(-6, 6, 3).astype(np.float32)
        put(k, c, c + rng.uniform(-1, 1, 3).astype(np.float32), c + rng.uniform(-1, 1, 3).astype(np.float32))
    s = ob.Scene("")
    s.set_materials(np.zeros(1, ob.MATERIAL))
    s.add_triangles(tris)
    s.build()
    mats, t2, nodes = s.arrays()
    assert nodes["count"].max() >= big_n and (nodes["count"] == 17).any()
    wf = pa.Wavefront()
    wf.upload_scene(mats, t2, nodes)
    n = 20000
    o = rng.uniform(-8, 8, (n, 3)).astype(np.float32)
    o[:, 2] = rng.uniform(2, 8, n)
    tgt = rng.uniform(-1, 7, (n, 3)).astype(np.float32)
    tgt[:, 2] = rng.uniform(-0.5, 1.5, n)
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([o, d.astype(np.float32)], 1)
    g = wf.trace_closest(rays)
    can = s.intersect(rays, "canonical")
    br = s.intersect(rays, "brute")
    lit = s.intersect(rays, "literal")
    assert np.array_equal(g["tri"], can["tri"]) and np.array_equal(bits(g["t"]), bits(can["t"]))
    assert np.array_equal(g["tri"], br["tri"]) and np.array_equal(bits(g["t"]), bits(lit["t"]))
    big = (t2["v1"]["position"][g["tri"].clip(0)] == base[0]).all(axis=1) & (g["tri"] >= 0)
    assert big.sum() > 100
    tmax = rng.uniform(0.5, 12, n).astype(np.float32)
    assert np.array_equal(wf.trace_occluded(rays, tmax), s.occluded(rays, tmax))
    wf.close()


def pa_assets():
    from conftest import MESHES
    return MESHES


def test_bad_bvh_rejected(pa, ob, bundled):
    mats, tris, nodes = bundled.arrays()
    wf = pa.Wavefront()
    bad = nodes.copy()
    bad["left_first"][0] = 2            # even child index
    with pytest.raises(pa.PtError) as e:
        wf.upload_scene(mats, tris, bad)
    assert e.value.code == -5
    bad = nodes.copy()
    leaf = np.nonzero(bad["count"] > 0)[0][0]
    bad["left_first"][leaf] = tris.size  # leaf range out of bounds
    with pytest.raises(pa.PtError):
        wf.upload_scene(mats, tris, bad)
    # two interior nodes that claim the same pair of children (and leave another pair to nobody)
    inner = np.nonzero(nodes["count"] <= 0)[0]
    bad = nodes.copy()
    bad["left_first"][inner[2]] = bad["left_first"][inner[1]] if bad["left_first"][inner[1]] > inner[2] else bad["left_first"][inner[3]]
    with pytest.raises(pa.PtError) as e:
        wf.upload_scene(mats, tris, bad)
    assert e.value.code == -5 and ("referenced" in str(e.value))
    # a leaf that forgets one of its triangles
    bad = nodes.copy()
    two = np.nonzero(bad["count"] >= 2)[0][0]
    bad["count"][two] -= 1
    with pytest.raises(pa.PtError) as e:
        wf.upload_scene(mats, tris, bad)
    assert e.value.code == -5 and "leaves cover" in str(e.value)
    # chains of sibling pairs: 40 levels are fine, 100 exceed the traversal stack, 140 exceed what the depth search resolves
    from pathtracer_amd.types import NODE
    for levels, ok in ((40, True), (100, False), (140, False)):
        chain = np.zeros(2 * levels + 1, NODE)
        chain["min"], chain["max"] = (-1, -1, -1), (1, 1, 1)
        for k in range(levels):                      # node 2k is interior (the root for k = 0): children 2k+1 (a leaf) and 2k+2
            chain["left_first"][2 * k], chain["count"][2 * k] = 2 * k + 1, -1
            chain["left_first"][2 * k + 1], chain["count"][2 * k + 1] = k, 1
        chain["left_first"][2 * levels], chain["count"][2 * levels] = levels, 1
        if ok:
            other = pa.Wavefront()
            other.upload_scene(mats, tris[:levels + 1], chain)
            other.close()
        else:
            with pytest.raises(pa.PtError) as e:
                wf.upload_scene(mats, tris[:levels + 1], chain)
            assert e.value.code == -5 and "depth" in str(e.value)
    badt = tris.copy()
    badt["material"][5] = 9999
    with pytest.raises(pa.PtError):
        wf.upload_scene(mats, badt, nodes)
    with pytest.raises(pa.PtError):
        wf.allocate(64, 64) or wf.invoke(bundled_view(bundled, 64, 64))   # no scene uploaded
    # a refused scene — by either path — leaves the installed one as it was
    wf.upload_scene(mats, tris, nodes)
    fs = bundled_view(bundled, 64, 64)
    wf.invoke(fs, spp=2, max_bounces=3)
    before = wf.read_image()
    with pytest.raises(pa.PtError):
        wf.upload_scene(mats, badt, nodes)
    with pytest.raises(pa.PtError):
        wf.build_scene(mats, badt)            # the GPU build notices the material index only after building
    fs["samples"] = 0
    wf.invoke(fs, spp=2, max_bounces=3)
    assert np.array_equal(bits(wf.read_image()), bits(before))
    wf.close()


def test_replacing_scenes_does_not_leak_device_memory(pa, bundled):
    """A host that re-uploads or rebuilds its scene over and over (key P in the reference, Game.cs:169-180) must get the memory of
    the scene it replaces back: 60 replacements by either entry point, refused ones among them, leave the device's free memory
    where it was."""
    import torch
    mats, tris, nodes, refs = bundled.arrays(references=True)
    wf = pa.Wavefront(nee=True, rng_mode=1)
    wf.upload_scene(mats, tris, nodes)
    wf.build_scene(mats, refs)
    wf.synchronize()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    badt = tris.copy()
    badt["material"][7] = 12345
    for k in range(20):
        wf.upload_scene(mats, tris, nodes)
        wf.build_scene(mats, refs)
        with pytest.raises(pa.PtError):
            (wf.upload_scene if k & 1 else wf.build_scene)(mats, badt, *(() if not (k & 1) else (nodes,)))
    wf.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < (8 << 20), f"{(free0 - free1) >> 20} MiB of device memory did not come back"
    wf.close()


def test_contexts_and_resizes_do_not_leak_device_memory(pa, bundled):
    """Contexts created and destroyed, and one context resized back and forth between frames (Game.Resize, Game.cs:138-151): the
    device's free memory ends where it began."""
    import torch
    mats, tris, nodes = bundled.arrays()

    def one_context(sizes):
        wf = pa.Wavefront(nee=True, rng_mode=1)
        wf.upload_scene(mats, tris, nodes)
        for w, h in sizes:
            wf.allocate(w, h)
            wf.invoke(bundled_view(bundled, w, h), spp=2, max_bounces=3)
            wf.read_image()
        wf.close()

    one_context([(64, 64)])
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(10):
        one_context([(64, 64)])
    one_context([(640, 360), (64, 48), (1280, 720), (90, 50), (640, 360)] * 3)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < (8 << 20), f"{(free0 - free1) >> 20} MiB of device memory did not come back"


def test_context_lifecycle(pa, bundled, cornell):
    """One context through the sequence a host goes through — wrong-order calls are refused, resizing and replacing
    the scene leave no stale state: every image equals a fresh context's."""
    def fresh(scene, w, h, spp, frames):
        x = make_wf(pa, scene, nee=True, rng_mode=1)
        x.allocate(w, h)
        f = bundled_view(scene, w, h, frames=frames)
        x.invoke(f, spp=spp, max_bounces=4)
        img = x.read_image()
        x.close()
        return img

    wf = pa.Wavefront(nee=True, rng_mode=1)
    fs = bundled_view(bundled, 64, 48, frames=1)
    with pytest.raises(pa.PtError) as e:
        wf.invoke(fs, spp=1, max_bounces=4)                    # no scene
    assert e.value.code == -6
    wf.upload_scene(*bundled.arrays())
    with pytest.raises(pa.PtError) as e:
        wf.invoke(fs, spp=1, max_bounces=4)                    # no image size
    assert e.value.code == -6
    with pytest.raises(pa.PtError):
        wf.allocate(0, 48)
    for key, bad in ((1, 3), (2, 2), (4, 3), (6, 17), (7, 0), (7, 65), (8, 0), (99, 1)):
        with pytest.raises(pa.PtError):
            wf.set_option(key, bad)
    wf.set_option(7, 16)                                        # scheduling knobs cannot change the image
    wf.set_option(8, 64)
    wf.allocate(64, 48)
    with pytest.raises(pa.PtError):
        wf.invoke(fs, spp=0, max_bounces=4)
    wf.invoke(fs, spp=2, max_bounces=4)
    assert np.array_equal(bits(wf.read_image()), bits(fresh(bundled, 64, 48, 2, 1)))
    wf.allocate(200, 104)                                       # larger: queues are reallocated, accumulation restarts
    fs = bundled_view(bundled, 200, 104, frames=3)
    wf.invoke(fs, spp=3, max_bounces=4)
    assert np.array_equal(bits(wf.read_image()), bits(fresh(bundled, 200, 104, 3, 3)))
    wf.upload_scene(*cornell.arrays())                          # another scene into the same context
    wf.reset_accumulation()
    fs = bundled_view(cornell, 200, 104, frames=1)
    wf.invoke(fs, spp=2, max_bounces=4)
    assert np.array_equal(bits(wf.read_image()), bits(fresh(cornell, 200, 104, 2, 1)))
    mats, tris, nodes, refs = bundled.arrays(references=True)   # ... and a scene built on the GPU into it
    wf.build_scene(mats, refs)
    wf.allocate(64, 48)
    fs = bundled_view(bundled, 64, 48, frames=1)
    wf.invoke(fs, spp=2, max_bounces=4)
    assert np.array_equal(bits(wf.read_image()), bits(fresh(bundled, 64, 48, 2, 1)))
    assert wf.stats()["stack_overflows"] == 0
    wf.close()


@pytest.mark.parametrize("nee,rng_mode,bounces", [(0, 0, 10), (1, 0, 4), (1, 1, 4), (0, 1, 2), (2, 1, 5)])
def test_radiance_bundled_256(pa, ob, bundled, wf_bundled, nee, rng_mode, bounces):
    """Bundled scene, every material type (diffuse/mirror/dielectric/metal), vs the oracle path tracer."""
    w, h, spp = 256, 256, 3
    wf = wf_bundled
    wf.set_option(1, nee)
    wf.set_option(2, rng_mode)
    wf.allocate(w, h)
    fs = bundled_view(bundled, w, h, frames=1)
    wf.reset_stats()
    wf.invoke(fs, spp=spp, max_bounces=bounces)
    img = wf.read_image()
    st = wf.stats()
    osum, ost = bundled.render(ob.render_state(w, h), fs, ob.path_config(bounces, nee, rng_mode), spp, threads=8)
    oimg = osum / np.float32(spp)
    assert st["stack_overflows"] == 0
    assert st["extend_rays"] == ost["extend_rays"] and st["shadow_rays"] == ost["shadow_rays"]
    ndiff = int((bits(img) != bits(oimg)).any(axis=2).sum())
    err = rel_rmse(img[..., :3], oimg[..., :3])
    print(f"nee={nee} mode={rng_mode}: relRMSE={err:.3e}, pixels differing bitwise={ndiff}/{w*h}, rays={st['extend_rays']}+{st['shadow_rays']}")
    assert np.isfinite(img).all() == np.isfinite(oimg).all()
    assert err <= RADIANCE_RELRMSE_TOL
    assert ndiff == 0
    wf.set_option(1, 0)
    wf.set_option(2, 0)


@pytest.mark.parametrize("w,h", [(100, 52), (13, 7), (8, 9)])
def test_radiance_ragged_image_sizes(pa, ob, bundled, wf_bundled, w, h):
    """Sides that are not multiples of the 8x8 pixel groups (the reference mis-handles them, generate.glsl:177-190; here
    the edge groups are partly empty): rays are compacted at generation, and the image still equals the oracle's."""
    wf = wf_bundled
    wf.set_option(1, 2)
    wf.set_option(2, 1)
    wf.allocate(w, h)
    fs = bundled_view(bundled, w, h, frames=2)
    wf.reset_stats()
    wf.invoke(fs, spp=3, max_bounces=4)
    img = wf.read_image()
    st = wf.stats()
    osum, ost = bundled.render(ob.render_state(w, h), fs, ob.path_config(4, 2, 1), 3, threads=8)
    assert st["extend_rays"] == ost["extend_rays"] and st["shadow_rays"] == ost["shadow_rays"]
    assert st["paths"] == 3 * w * h
    assert np.array_equal(bits(img), bits(osum / np.float32(3)))
    wf.set_option(1, 0)
    wf.set_option(2, 0)


@pytest.mark.parametrize("nee", [1, 2])
def test_radiance_cornell_nee(pa, ob, cornell, nee):
    """Cornell box (quads, negative indices, no normals -> flat-normal fallback), NEE on, without and with multiple
    importance sampling."""
    w, h, spp = 128, 128, 4
    wf = make_wf(pa, cornell, nee=nee, rng_mode=1)
    wf.allocate(w, h, sky=(0, 0, 0, 1))
    fs, _ = cornell.camera((0, 1, 3.4), (0, 1, 0), w, h, use_scene=True, frames=1, samples=1)
    wf.invoke(fs, spp=spp, max_bounces=5)
    img = wf.read_image()
    osum, _ = cornell.render(ob.render_state(w, h, sky=(0, 0, 0, 1)), fs, ob.path_config(5, nee, 1), spp, threads=8)
    oimg = osum / np.float32(spp)
    assert oimg[..., :3].mean() > 0.01
    finite = np.isfinite(oimg).all(axis=2)
    assert np.array_equal(np.isfinite(img).all(axis=2), finite)
    assert rel_rmse(img[finite][:, :3], oimg[finite][:, :3]) <= RADIANCE_RELRMSE_TOL
    assert np.array_equal(bits(img)[finite], bits(oimg)[finite])
    wf.close()


def test_three_estimators_have_one_expectation(pa, cornell):
    """The reference's estimator (emitters counted when a path runs into them, shade.glsl:366-370), next-event
    estimation, and the two combined by multiple importance sampling must converge to the same image.  Cornell box,
    64x64, 65,536 samples per pixel each, compared on 8x8-pixel blocks.  Tolerances (stated here, statistical): block
    median of the relative difference <= 1.5 %, whole-image mean <= 1 %; plain light sampling is left out of the mean
    check because next to the two-sided lamp its 1/distance^2 weights have a heavy tail that 64k samples do not average
    out (which is what the combination is for)."""
    w = h = 64
    spp = 65536
    fs, _ = cornell.camera((0, 1, 3.4), (0, 1, 0), w, h, use_scene=True, frames=1, samples=1)
    imgs = {}
    for nee in (0, 1, 2):
        wf = make_wf(pa, cornell, nee=nee, rng_mode=1)
        wf.allocate(w, h, sky=(0, 0, 0, 1))
        wf.invoke(fs, spp=spp, max_bounces=8)
        imgs[nee] = wf.read_image()[..., :3].astype(np.float64)
        assert np.isfinite(imgs[nee]).all()
        wf.close()
    blocks = {k: v.reshape(h // 8, 8, w // 8, 8, 3).mean(axis=(1, 3, 4)) for k, v in imgs.items()}
    for a, b in ((0, 1), (0, 2), (1, 2)):
        rel = np.abs(blocks[a] - blocks[b]) / np.maximum(blocks[a], 1e-6)
        print(f"estimators {a} vs {b}: block median {np.median(rel):.4f}, p90 {np.percentile(rel, 90):.4f}, "
              f"means {imgs[a].mean():.5f} {imgs[b].mean():.5f}")
        assert np.median(rel) <= 0.015
    assert abs(imgs[0].mean() - imgs[2].mean()) <= 0.01 * imgs[0].mean()


def test_long_paths_in_a_closed_box(pa, ob, cornell):
    """64 segments per path without next-event estimation (the reference's estimator) in the Cornell box: paths end by
    roulette or on the lamp, the queues ping-pong 64 times, and the image still equals the oracle's."""
    w, h, spp = 48, 48, 3
    wf = make_wf(pa, cornell, nee=0, rng_mode=1)
    wf.allocate(w, h, sky=(0, 0, 0, 1))
    fs, _ = cornell.camera((0, 1, 3.4), (0, 1, 0), w, h, use_scene=True, frames=5, samples=5)
    wf.invoke(fs, spp=spp, max_bounces=64)
    img = wf.read_image()
    st = wf.stats()
    osum, ost = cornell.render(ob.render_state(w, h, sky=(0, 0, 0, 1)), fs, ob.path_config(64, 0, 1), spp, threads=8)
    assert st["extend_rays"] == ost["extend_rays"] and st["extend_rays"] > 3 * spp * w * h // 2
    finite = np.isfinite(osum).all(axis=2)
    assert np.array_equal(np.isfinite(img).all(axis=2), finite)
    assert np.array_equal(bits(img)[finite], bits(osum / np.float32(spp))[finite])
    with pytest.raises(pa.PtError):
        wf.invoke(fs, spp=1, max_bounces=65)
    wf.close()


def test_traversal_counters_match_oracle(pa, ob, bundled):
    w, h = 128, 128
    wf = make_wf(pa, bundled, traversal_stats=True, nee=True)
    wf.allocate(w, h)
    fs = bundled_view(bundled, w, h, frames=2)
    wf.invoke(fs, spp=1, max_bounces=3)
    st = wf.stats()
    _, ost = bundled.render(ob.render_state(w, h), fs, ob.path_config(3, 1, 0), 1, threads=4)
    for k in ["extend_rays", "shadow_rays", "extend_steps", "extend_tris", "shadow_steps", "shadow_tris"]:
        assert st[k] == ost[k], k
    wf.close()


@pytest.mark.parametrize("w,h", [(96, 64), (90, 50)])
def test_batch_size_and_accumulation_invariance(pa, bundled, w, h):
    """The image must not depend on how many samples are traced concurrently, nor on call splitting (whole 8x8 groups, and a
    ragged size whose batches take the cleared-plane path)."""
    imgs = []
    for batch, calls in [(1, [4]), (4, [4]), (3, [4]), (2, [1, 3])]:
        wf = make_wf(pa, bundled, nee=True, rng_mode=1, samples_per_batch=batch)
        wf.allocate(w, h)
        fs = bundled_view(bundled, w, h, frames=1)
        for spp in calls:
            wf.invoke(fs, spp=spp, max_bounces=4)
            fs["frames"] += spp
            fs["samples"] += spp
        imgs.append(wf.read_image())
        wf.close()
    for im in imgs[1:]:
        assert np.array_equal(bits(im), bits(imgs[0]))


def test_overlapping_render_calls(pa, ob, bundled):
    """Consecutive pt_render calls are traced on two alternating sets of queues and overlap on the device; their samples enter the
    accumulator in call order on the context's stream.  A run of calls queued without waiting — single- and multi-batch, a reset of
    the accumulation in the middle, a read-out and a closest-hit query between calls, a resize — must give the bits and the
    counters of the same run with the second set switched off (PT_OPT_PIPELINE_CALLS = 0), and the first frame those of the oracle."""
    w, h = 160, 96
    got = {}
    # (second set on, samples per batch, counting kernels, quantised node records): the kernels a call that shares the GPU launches
    # (small shade blocks, connect inside the next extend) exist with and without counters and for both kinds of records
    runs = [(0, 0, True, 0), (0, 1, True, 0), (1, 0, True, 0), (1, 1, True, 0), (1, 0, False, 0), (1, 1, False, 0),
            (0, 0, True, 1), (1, 0, True, 1), (1, 1, False, 1)]
    for pipe, batch, stats, quant in runs:
        wf = make_wf(pa, bundled, nee=True, rng_mode=1, traversal_stats=stats, samples_per_batch=batch)
        wf.set_option(11, pipe)
        wf.set_option(9, quant)
        wf.allocate(w, h)
        fs = bundled_view(bundled, w, h, frames=1)
        fs["samples"] = 0
        f = fs.copy()
        imgs = []
        for k, spp in enumerate((2, 3, 1, 2, 4)):
            wf.invoke(f, spp=spp, max_bounces=4, wait=False)
            f["frames"] += spp
            f["samples"] += spp
            if k == 0:
                imgs.append(wf.read_image())                       # between two calls in flight
            if k == 2:
                rays = np.array([[0, 2, 15, 0.01, 0.05, -1.0]], np.float32)
                assert wf.trace_closest(rays)["tri"][0] >= 0       # a query on the context's stream, which shares the first set's control block
        imgs.append(wf.read_image())
        st = wf.stats()
        counters = tuple(st[k] for k in (("paths", "extend_rays", "shadow_rays", "extend_steps", "extend_tris", "shadow_steps", "shadow_tris") if stats
                                         else ("paths", "extend_rays", "shadow_rays")))
        assert st["stack_overflows"] == 0
        f["samples"] = 0                                           # FrameState.Samples == 0: the accumulation starts over, two calls deep
        wf.invoke(f, spp=2, max_bounces=4, wait=False)
        f["frames"] += 2
        f["samples"] += 2
        wf.invoke(f, spp=2, max_bounces=4, wait=False)
        imgs.append(wf.read_image())
        wf.allocate(96, 64)                                        # a resize with nothing waited for
        wf.invoke(bundled_view(bundled, 96, 64, frames=3), spp=3, max_bounces=4, wait=False)
        wf.invoke(bundled_view(bundled, 96, 64, frames=6), spp=1, max_bounces=4, wait=False)
        imgs.append(wf.read_image())
        got[(pipe, batch, stats, quant)] = (imgs, counters)
        wf.close()
    ref_imgs, _ = got[(0, 0, True, 0)]
    for key, (imgs, counters) in got.items():
        # the counters of one kind of records agree among themselves (the quantised walk takes a few more steps); rays always do
        ref_counters = got[(0, 0, True, key[3])][1]
        assert counters == (ref_counters if key[2] else ref_counters[:3]), key
        for a, b in zip(imgs, ref_imgs):
            assert np.array_equal(bits(a), bits(b)), key
    fs = bundled_view(bundled, w, h, frames=1)
    osum, _ = bundled.render(ob.render_state(w, h), fs, ob.path_config(4, 1, 1), 2)
    assert np.array_equal(bits(ref_imgs[0]), bits(osum / np.float32(2)))


def test_full_size_frame_properties(pa, ob, bundled):
    """The headline configuration at full size (1920x1080, 16 spp, 4 bounces, NEE), through properties that need no
    CPU render: the image is the same bits for 16 samples in one batch, in batches of 5, and as 4 + 12 over two calls;
    it is finite, non-negative, and its ray counts are the same in all three runs."""
    w, h = 1920, 1080
    imgs, rays = [], []
    for batch, calls in [(16, [16]), (5, [16]), (0, [4, 12])]:
        wf = make_wf(pa, bundled, nee=True, rng_mode=1, samples_per_batch=batch)
        wf.allocate(w, h)
        fs = bundled_view(bundled, w, h, frames=1)
        for spp in calls:
            wf.invoke(fs, spp=spp, max_bounces=4)
            fs["frames"] += spp
            fs["samples"] += spp
        imgs.append(wf.read_image())
        st = wf.stats()
        rays.append((st["paths"], st["extend_rays"], st["shadow_rays"]))
        assert st["stack_overflows"] == 0
        wf.close()
    assert rays[0][0] == w * h * 16 and rays[0] == rays[1] == rays[2]
    assert np.isfinite(imgs[0]).all() and (imgs[0] >= 0).all() and imgs[0][..., :3].mean() > 0.05
    assert np.array_equal(bits(imgs[1]), bits(imgs[0])) and np.array_equal(bits(imgs[2]), bits(imgs[0]))


def test_full_size_frames_in_flight(pa, bundled):
    """Three headline frames (1920x1080, 16 spp, 4 bounces, NEE) queued back to back, so that two are in flight and the calls take
    the kernels of frames that share the GPU (four traversal blocks per CU, 256-thread shade blocks, connect inside the next extend
    launch): the accumulated image and the ray counts are those of the same three calls with one set of queues, bit for bit."""
    w, h = 1920, 1080
    got = []
    for pipe in (1, 0):
        wf = make_wf(pa, bundled, nee=True, rng_mode=1)
        wf.set_option(11, pipe)
        wf.allocate(w, h)
        fs = bundled_view(bundled, w, h, frames=1)
        fs["samples"] = 0
        for _ in range(3):
            wf.invoke(fs, spp=16, max_bounces=4, wait=False)
            fs["frames"] += 16
            fs["samples"] += 16
        img = wf.read_image()
        st = wf.stats()
        assert st["stack_overflows"] == 0
        got.append((img, (st["paths"], st["extend_rays"], st["shadow_rays"])))
        wf.close()
    assert got[0][1][0] == w * h * 48 and got[0][1] == got[1][1]
    assert np.array_equal(bits(got[0][0]), bits(got[1][0]))


def test_million_triangle_soup_against_the_oracle(pa, ob):
    """The 1M-triangle configuration's scene at full size: GPU build == oracle build, and incoherent closest / any hits
    == the oracle's literal and canonical walks on a sample of rays."""
    o = ob.Scene("")
    o.soup(1000000)
    o.build()
    mats, tris, nodes, refs = o.arrays(references=True)
    wf = pa.Wavefront()
    got_t, got_n, _ = wf.build_scene(mats, refs, want_arrays=True)
    assert got_n.tobytes() == nodes.tobytes() and got_t.tobytes() == tris.tobytes()
    rng = np.random.default_rng(17)
    n = 8192
    rays = np.zeros((n, 6), np.float32)
    rays[:, 0:3] = rng.uniform(-50, 50, (n, 3))
    d = rng.normal(size=(n, 3)).astype(np.float32)
    rays[:, 3:6] = d / np.sqrt((d * d).sum(1, dtype=np.float32))[:, None]
    g = wf.trace_closest(rays)
    can, lit = o.intersect(rays, "canonical", threads=8), o.intersect(rays, "literal", threads=8)
    assert np.array_equal(g["tri"], can["tri"]) and np.array_equal(bits(g["t"]), bits(can["t"])) and np.array_equal(bits(g["uv"]), bits(can["uv"]))
    assert np.array_equal(g["tri"], lit["tri"]) and np.array_equal(bits(g["t"]), bits(lit["t"]))
    assert (g["tri"] >= 0).mean() > 0.5
    tmax = rng.uniform(1, 60, n).astype(np.float32)
    assert np.array_equal(wf.trace_occluded(rays, tmax), o.occluded(rays, tmax, threads=8))
    wf.close()


def test_plain_c_host_renders_the_same_image(pa, bundled, tmp_path):
    """examples/render_c (plain C over pt_host.h + pt_hip.h, no Python in the process) against the ctypes path: same
    bytes, with the CPU-built upload and with the GPU build + repack."""
    import subprocess
    from conftest import MESHES, ROOT
    w, h, spp, bounces = 160, 96, 3, 4
    wf = make_wf(pa, bundled, nee=True, rng_mode=1)
    wf.allocate(w, h)
    fs = bundled_view(bundled, w, h, frames=1)
    fs["samples"] = 0
    wf.invoke(fs, spp=spp, max_bounces=bounces)
    want = wf.read_image()
    wf.close()
    exe = os.path.join(ROOT, "examples", "render_c")
    if not os.path.exists(exe):   # built by __graft_entry__.build(); a tree that lost the binary builds it here
        exe = str(tmp_path / "render_c")
        pkg = os.path.join(ROOT, "pathtracer_amd")
        subprocess.run(["gcc", "-std=c99", "-O1", "-I" + os.path.join(ROOT, "include"), "-o", exe, os.path.join(ROOT, "examples", "render_c.c"),
                        "-L" + pkg, "-lpt_hip", "-lpt_host", "-Wl,-rpath," + pkg, "-Wl,--allow-shlib-undefined"], check=True)
    for extra in ([], ["gpu-build"]):
        out = tmp_path / "img.raw"
        r = subprocess.run([exe, MESHES, str(w), str(h), str(spp), str(bounces), str(out)] + extra,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        got = np.fromfile(out, np.float32).reshape(h, w, 4)
        if not extra:
            assert np.array_equal(bits(got), bits(want)), r.stdout
        else:
            # a different insertion order gives a different (valid) tree and light-table order: another estimate of
            # the same picture
            assert np.isfinite(got).all() and abs(float(got[..., :3].mean()) / float(want[..., :3].mean()) - 1.0) < 0.03, r.stdout


def test_one_context_over_several_gpus(pa, ob, bundled, tmp_path):
    """pt_config.device_count: one context, the same calls, N member contexts inside the library.  Rehearsed on the one GPU
    there is (PT_GROUP_SAME_DEVICE: three members on device 0, tiles gathered by device copies) — the image, the ray counts
    and the traversal counters equal a single-GPU context's bit for bit, through ctypes and through the plain-C host; a
    ragged image (77 groups over 3 members), the GPU-built scene and a second accumulation are covered too.  The RCCL path
    itself (library loading, communicator, grouped ncclSend / ncclRecv on the members' streams) runs with the one rank a
    single GPU allows (PT_GROUP_FORCE_RCCL)."""
    import subprocess
    from conftest import MESHES, ROOT
    mats, tris, nodes = bundled.arrays()
    for (w, h) in [(160, 96), (88, 56)]:
        fs = bundled_view(bundled, w, h, frames=1)
        fs["samples"] = 0
        imgs, stats = {}, {}
        for name, kw in [("one", {}), ("three", dict(device_count=3, group_flags=1)), ("rccl", dict(device_count=1, group_flags=2))]:
            wf = pa.Wavefront(nee=True, rng_mode=1, traversal_stats=True, **kw)
            wf.upload_scene(mats, tris, nodes)
            wf.allocate(w, h)
            f = fs.copy()
            for spp in (2, 3):                      # two calls: accumulation carries over inside every member
                wf.invoke(f, spp=spp, max_bounces=4)
                f["frames"] += spp
                f["samples"] += spp
            imgs[name] = wf.read_image()
            st = wf.stats()
            stats[name] = tuple(st[k] for k in ("paths", "extend_rays", "shadow_rays", "extend_steps", "extend_tris", "shadow_steps", "shadow_tris"))
            assert st["stack_overflows"] == 0
            if name == "three":
                rays = np.array([[0, 2, 15, 0.01, 0.05, -1.0]], np.float32)
                assert wf.trace_closest(rays)["tri"][0] >= 0
                with pytest.raises(pa.PtError):
                    wf.accum_device()
                wf.invoke(fs, spp=2, max_bounces=4)     # FrameState.Samples == 0: the accumulation restarts by itself
                again = wf.read_image()
            wf.close()
        assert np.array_equal(bits(imgs["three"]), bits(imgs["one"])) and np.array_equal(bits(imgs["rccl"]), bits(imgs["one"]))
        assert stats["three"] == stats["one"] == stats["rccl"]
        ref = pa.Wavefront(nee=True, rng_mode=1)
        ref.upload_scene(mats, tris, nodes)
        ref.allocate(w, h)
        ref.invoke(fs, spp=2, max_bounces=4)
        assert np.array_equal(bits(again), bits(ref.read_image()))
        ref.close()
    # GPU build on every member
    from pathtracer_amd.host import Scene
    s = Scene()
    s.initialize_scene(11)
    hm, refs = s.references()
    got = []
    for kw in ({}, dict(device_count=2, group_flags=1)):
        wf = pa.Wavefront(nee=True, rng_mode=1, **kw)
        wf.build_scene(hm, refs)
        wf.allocate(96, 64)
        wf.invoke(bundled_view(bundled, 96, 64, frames=1), spp=2, max_bounces=4)
        got.append(wf.read_image())
        wf.close()
    assert np.array_equal(bits(got[0]), bits(got[1]))
    # the plain-C host with devices=3
    exe = os.path.join(ROOT, "examples", "render_c")
    outs = []
    for extra in ([], ["devices=3", "same-device"], ["force-rccl"]):
        out = tmp_path / "img.raw"
        r = subprocess.run([exe, MESHES, "160", "96", "3", "4", str(out)] + extra, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append(np.fromfile(out, np.float32))
    assert np.array_equal(bits(outs[0]), bits(outs[1])) and np.array_equal(bits(outs[0]), bits(outs[2]))
    with pytest.raises(pa.PtError):
        pa.Wavefront(device_count=2)            # two distinct GPUs asked for, one present


def test_command_line_renderer(pa, bundled, tmp_path):
    """python -m pathtracer_amd: the float image it writes equals a render through the API; the GPU-built variant (another
    valid tree over the same triangles) gives the same picture up to its light order; the PNG is well formed."""
    import struct
    import subprocess
    import sys
    import zlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = [sys.executable, "-m", "pathtracer_amd", "--width", "160", "--height", "96", "--spp", "4", "--bounces", "4", "--nee", "1"]

    def run(extra):
        r = subprocess.run(common + extra, cwd=root, capture_output=True, text=True, timeout=240)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "165208 triangles" in r.stdout
        return r.stdout

    def read_pfm(path):
        b = open(path, "rb").read()
        head = b"PF\n160 96\n-1.0\n"
        assert b.startswith(head)
        return np.frombuffer(b[len(head):], np.float32).reshape(96, 160, 3)

    run(["--out", str(tmp_path / "a.pfm")])
    run(["--out", str(tmp_path / "b.pfm"), "--gpu-build"])
    run(["--out", str(tmp_path / "a.png")])
    wf = make_wf(pa, bundled, nee=1, rng_mode=1)
    wf.allocate(160, 96)
    wf.invoke(bundled_view(bundled, 160, 96, frames=1), spp=4, max_bounces=4)
    ref = wf.read_image()[..., :3]
    wf.close()
    a, b = read_pfm(tmp_path / "a.pfm"), read_pfm(tmp_path / "b.pfm")
    assert np.array_equal(bits(a), bits(ref))
    assert np.isfinite(b).all() and abs(float(b.mean()) - float(a.mean())) <= 0.03 * float(a.mean())
    png = open(tmp_path / "a.png", "rb").read()
    assert png[:8] == b"\x89PNG\r\n\x1a\n"
    w_, h_ = struct.unpack(">II", png[16:24])
    assert (w_, h_) == (160, 96)
    idat = png[png.index(b"IDAT") + 4:png.index(b"IEND") - 8]
    assert len(zlib.decompress(idat)) == 96 * (160 * 3 + 1)


@pytest.mark.parametrize("w,h", [(88, 56), (75, 41)])   # 11 x 7 = 77 groups: not divisible by 3; 10 x 6 groups with ragged edges
def test_tile_sharding_single_process(pa, ob, bundled, w, h):
    """world=3 contexts on one GPU: assembling their accumulators gives the 1-GPU image bit for bit."""
    import torch
    spp = 2
    fs = bundled_view(bundled, w, h, frames=1)
    ref = make_wf(pa, bundled, nee=True)
    ref.allocate(w, h)
    ref.invoke(fs, spp=spp, max_bounces=3)
    want = ref.read_image()
    ref.close()
    world = 3
    parts, ctxs = [], []
    for r in range(world):
        c = make_wf(pa, bundled, nee=True, tile_rank=r, tile_world=world)
        c.allocate(w, h)
        c.invoke(fs, spp=spp, max_bounces=3)
        ptr, n, s = c.accum_device()
        assert s == spp
        parts.append((ptr, n))
        ctxs.append(c)
    n = parts[0][1]
    assert all(p[1] == n for p in parts)
    from pathtracer_amd.tiles import accumulator_tensor
    gathered = torch.empty((world, n, 4), dtype=torch.float32, device="cuda")
    for r, c in enumerate(ctxs):
        gathered[r].copy_(accumulator_tensor(c))
    torch.cuda.synchronize()
    got = ctxs[0].assemble_tiles(gathered.data_ptr(), world, spp)
    assert np.array_equal(bits(got), bits(want))
    for c in ctxs:
        c.close()


RANK_WORKER = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, {root!r})
import pathtracer_amd as pa
from pathtracer_amd import tiles
from pathtracer_amd.host import Scene, Camera

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
w, h, spp = 200, 120, 3
s = Scene(); s.initialize_scene(11); s.build(2)
fs = Camera(w, h).frame_state((0, 2, 15), (0, 3, 0), frames=1, samples=0)
wf = pa.Wavefront(device=0, nee=True, rng_mode=1, tile_rank=rank, tile_world=world)
wf.upload_scene(*s.arrays())
wf.allocate(w, h)
wf.invoke(fs, spp=spp, max_bounces=4)
img = tiles.gather_image(wf, spp, dst=0)
if rank == 0:
    ref = pa.Wavefront(device=0, nee=True, rng_mode=1)
    ref.upload_scene(*s.arrays())
    ref.allocate(w, h)
    ref.invoke(fs, spp=spp, max_bounces=4)
    want = ref.read_image()
    assert np.array_equal(img.view(np.uint32), want.view(np.uint32)), "tile-sharded image differs from the single-context image"
else:
    assert img is None
# a second image through the preallocated gather bench.py uses
fs["frames"] += spp
fs["samples"] += spp
gather = tiles.AccumulatorGather(wf, dst=0)
wf.invoke(fs, spp=spp, max_bounces=4, wait=False)
res = gather()
torch.cuda.synchronize()
if rank == 0:
    dev = res.to("cuda")
    torch.cuda.synchronize()
    img2 = wf.assemble_tiles(dev.data_ptr(), world, 2 * spp)
    ref.invoke(fs, spp=spp, max_bounces=4)
    want2 = ref.read_image()
    assert np.array_equal(img2.view(np.uint32), want2.view(np.uint32)), "second gathered image differs"
    print("RANKS_OK")
else:
    assert res is None
dist.barrier()
wf.close()
dist.destroy_process_group()
"""


def test_two_processes_share_one_gpu(tmp_path):
    """One process per rank (as bench.py --gpus N runs), both on cuda:0, gathered over gloo: the assembled image
    equals the single-context image bit for bit."""
    import subprocess
    import sys
    from conftest import ROOT
    script = tmp_path / "rank_worker.py"
    script.write_text(RANK_WORKER.format(root=ROOT))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29617", str(script)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "RANKS_OK" in r.stdout


NCCL_WORKER = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, {root!r})
import pathtracer_amd as pa
from pathtracer_amd import tiles
from pathtracer_amd.host import Scene, Camera

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
w, h, spp = 200, 120, 3
s = Scene(); s.initialize_scene(11); s.build(2)
fs = Camera(w, h).frame_state((0, 2, 15), (0, 3, 0), frames=1, samples=0)
wf = pa.Wavefront(device=0, nee=True, rng_mode=1, tile_rank=0, tile_world=1)
wf.upload_scene(*s.arrays())
wf.allocate(w, h)
gather = tiles.AccumulatorGather(wf, dst=0)
assert gather.on_device
for k in range(2):                      # two images: the second gather is ordered after the first by stream semantics
    wf.invoke(fs, spp=spp, max_bounces=4, wait=False)
    fs["frames"] += spp
    fs["samples"] += spp
    res = gather()
torch.cuda.synchronize()
assert res.is_cuda and tuple(res.shape[:1]) == (1,)
img = wf.assemble_tiles(res.data_ptr(), 1, 2 * spp)
want = wf.read_image()
assert np.array_equal(img.view(np.uint32), want.view(np.uint32))
# the gathered image must be image k even when the caller overwrites the accumulator straight after gather():
# a reset and a different frame are enqueued without any wait in between
wf.reset_accumulation()
wf.invoke(fs, spp=spp, max_bounces=4)
want_a = wf.read_image()
wf.reset_accumulation()
wf.invoke(fs, spp=spp, max_bounces=4, wait=False)
res = gather()
wf.reset_accumulation()
fs2 = Camera(w, h).frame_state((3, 4, 12), (0, 1, 0), frames=50, samples=0)
for _ in range(4):
    wf.invoke(fs2, spp=8, max_bounces=4, wait=False)
torch.cuda.synchronize()
wf.synchronize()
img_a = wf.assemble_tiles(res.data_ptr(), 1, spp)
assert np.array_equal(img_a.view(np.uint32), want_a.view(np.uint32)), "the gathered tiles hold samples of the next frame"
t = torch.ones(4, device="cuda")
dist.all_reduce(t)
dist.barrier()
wf.close()
dist.destroy_process_group()
print("NCCL_OK")
"""


def test_rccl_single_rank_gather(tmp_path):
    """The RCCL ("nccl") code path of the tile gather on the one GPU there is: process group, device-resident snapshot,
    gather on RCCL's stream, assembly from the gathered tensor.  (More ranks need more GPUs; the partition logic is
    covered by the gloo tests.)"""
    import subprocess
    import sys
    from conftest import ROOT
    script = tmp_path / "nccl_worker.py"
    script.write_text(NCCL_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29633", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "NCCL_OK" in r.stdout
