"""GPU parity on the benchmark configurations the bundled scene does not cover: the Sponza-scale scene at 3840x2160
(configuration 4), the 20M-triangle instanced scene (configuration 5), and a full render of the 1M-triangle soup
(configuration 3) against the oracle.  Same bars as test_gpu_parity.py: ids, distances, flags, images and counters
bit for bit; full sizes through properties that need no CPU render.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def same_bytes(a, b):
    return a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8))


@pytest.fixture(scope="module")
def pa():
    import pathtracer_amd
    return pathtracer_amd


def incoherent_rays(rng, n, lo, hi):
    rays = np.zeros((n, 6), np.float32)
    rays[:, 0:3] = rng.uniform(lo, hi, (n, 3))
    d = rng.normal(size=(n, 3)).astype(np.float32)
    rays[:, 3:6] = d / np.sqrt((d * d).sum(1, dtype=np.float32))[:, None]
    return rays


def test_config4_sponza_scale_scene(pa, ob):
    """Bundled recipe + 19 bunnies (259,600 triangles) at 3840x2160: the GPU build equals the oracle's literal
    Accelerator.Build byte for byte; the primary hits of the central 512x512 window of the 4K frame equal the literal C#
    traversal with ZERO differing ids; at full size the image is the same bits however the samples are batched."""
    from pathtracer_amd.host import Scene
    s = Scene()
    s.initialize_scene_extended(11, 19)
    mats, refs = s.references()
    assert refs.size == 259600
    o = ob.Scene("")
    o.set_materials(mats)
    o.add_triangles(refs)
    o.build()                                   # the literal one-thread build
    _, otris, onodes = o.arrays()
    wf = pa.Wavefront(nee=True, rng_mode=1)
    got_t, got_n, _ = wf.build_scene(mats, refs, want_arrays=True)
    assert same_bytes(got_n, onodes) and same_bytes(got_t, otris)

    w, h = 3840, 2160
    fs, _ = o.camera((0, 2, 15), (0, 3, 0), w, h, use_scene=False, frames=1, samples=1)
    rays, pix = ob.generate(ob.render_state(w, h), fs, ob.path_config(), w * h)
    x, y = pix % w, pix // w
    window = (x >= 1664) & (x < 2176) & (y >= 824) & (y < 1336)
    assert int(window.sum()) == 512 * 512
    rays = np.ascontiguousarray(rays[window])
    g = wf.trace_closest(rays)
    lit = o.intersect(rays, "literal", threads=8)
    assert (lit["tri"] >= 0).mean() > 0.3
    assert int((g["tri"] != lit["tri"]).sum()) == 0
    assert np.array_equal(bits(g["t"]), bits(lit["t"]))

    imgs, counts = [], []
    for batch, calls in [(4, [4]), (3, [4]), (0, [1, 3])]:
        wf.set_option(5, batch)
        wf.allocate(w, h)
        wf.reset_stats()
        f = fs.copy()
        for spp in calls:
            wf.invoke(f, spp=spp, max_bounces=4)
            f["frames"] += spp
            f["samples"] += spp
        imgs.append(wf.read_image())
        st = wf.stats()
        counts.append((st["paths"], st["extend_rays"], st["shadow_rays"]))
        assert st["stack_overflows"] == 0
    assert counts[0][0] == 4 * w * h and counts[0] == counts[1] == counts[2]
    assert np.isfinite(imgs[0]).all() and (imgs[0] >= 0).all() and imgs[0][..., :3].mean() > 0.05
    assert np.array_equal(bits(imgs[1]), bits(imgs[0])) and np.array_equal(bits(imgs[2]), bits(imgs[0]))
    wf.close()


def test_config5_instanced_2m_against_the_literal_build(pa, ob):
    """The instanced-bunny generator at a size the oracle's one-thread literal build finishes in seconds (403 instances,
    2.0M triangles): GPU build == literal build, closest and any hits == the literal and canonical walks."""
    from pathtracer_amd.host import Scene
    s = Scene()
    s.generate_instances(403, 11)
    mats, refs = s.references()
    o = ob.Scene("")
    o.set_materials(mats)
    o.add_triangles(refs)
    o.build()
    _, otris, onodes = o.arrays()
    wf = pa.Wavefront()
    got_t, got_n, _ = wf.build_scene(mats, refs, want_arrays=True)
    assert same_bytes(got_n, onodes) and same_bytes(got_t, otris)
    lo, hi = onodes[0]["min"], onodes[0]["max"]
    rays = incoherent_rays(np.random.default_rng(3), 8192, lo, hi)
    g = wf.trace_closest(rays)
    can, lit = o.intersect(rays, "canonical", threads=8), o.intersect(rays, "literal", threads=8)
    assert (g["tri"] >= 0).mean() > 0.2
    assert np.array_equal(g["tri"], can["tri"]) and np.array_equal(bits(g["t"]), bits(can["t"])) and np.array_equal(bits(g["uv"]), bits(can["uv"]))
    assert np.array_equal(bits(g["t"]), bits(lit["t"]))
    differs = np.nonzero(g["tri"] != lit["tri"])[0]          # only exact distance ties may differ, towards the smaller index
    assert (g["tri"][differs] < lit["tri"][differs]).all() and len(differs) <= 8
    tmax = np.random.default_rng(4).uniform(1, 40, 8192).astype(np.float32)
    assert np.array_equal(wf.trace_occluded(rays, tmax), o.occluded(rays, tmax, threads=8))
    assert wf.stats()["stack_overflows"] == 0
    wf.close()


def test_config5_twenty_million_triangles(pa, ob):
    """The full 20.0M-triangle scene (4,026 instances): the GPU build equals the multi-threaded host mirror of
    Accelerator.Build byte for byte (the oracle's one-thread literal build of 20M triangles does not fit a test's time;
    the 2M-triangle test above ties the same generator and builders to it), the scene is render-ready straight from the
    build, and 8,192 incoherent closest / any hits equal the oracle's literal and canonical walks over those arrays."""
    from pathtracer_amd.host import Scene
    s = Scene()
    s.generate_instances(4026, 11)
    mats, refs = s.references()
    assert refs.size == 4026 * 4968 + 2
    wf = pa.Wavefront(nee=True, rng_mode=1)
    got_t, got_n, ms = wf.build_scene(mats, refs, want_arrays=True)
    s.build(16)
    _, htris, hnodes = s.arrays()
    assert same_bytes(got_n, hnodes)
    assert same_bytes(got_t, htris)
    print(f"20M triangles: {got_n.size} nodes, GPU build + repack {ms:.0f} ms")
    o = ob.Scene("")
    o.set_built(mats, got_t, got_n)               # the oracle walks the arrays; it did not build them
    del htris, hnodes, s
    lo, hi = got_n[0]["min"], got_n[0]["max"]
    rays = incoherent_rays(np.random.default_rng(7), 8192, lo, hi)
    g = wf.trace_closest(rays)
    can, lit = o.intersect(rays, "canonical", threads=8), o.intersect(rays, "literal", threads=8)
    assert (g["tri"] >= 0).mean() > 0.2
    assert np.array_equal(g["tri"], can["tri"]) and np.array_equal(bits(g["t"]), bits(can["t"])) and np.array_equal(bits(g["uv"]), bits(can["uv"]))
    assert np.array_equal(bits(g["t"]), bits(lit["t"]))
    differs = np.nonzero(g["tri"] != lit["tri"])[0]
    assert (g["tri"][differs] < lit["tri"][differs]).all() and len(differs) <= 8
    tmax = np.random.default_rng(8).uniform(1, 60, 8192).astype(np.float32)
    assert np.array_equal(wf.trace_occluded(rays, tmax), o.occluded(rays, tmax, threads=8))
    # render-ready: a small frame of the configuration's view, 8 bounces, against the oracle over the same arrays
    w, h, spp = 128, 72, 2
    wf.allocate(w, h)
    fs, _ = o.camera((0, 25, 70), (0, 18, 0), w, h, use_scene=False, frames=1, samples=1)
    wf.reset_stats()
    wf.invoke(fs, spp=spp, max_bounces=8)
    img = wf.read_image()
    st = wf.stats()
    osum, ost = o.render(ob.render_state(w, h), fs, ob.path_config(8, 1, 1), spp, threads=8)
    assert st["stack_overflows"] == 0
    assert st["extend_rays"] == ost["extend_rays"] and st["shadow_rays"] == ost["shadow_rays"]
    assert np.array_equal(bits(img), bits(osum / np.float32(spp)))
    # at the configuration's full size (3840x2160, 8 bounces; 3 of its 1024 spp): same bits however the samples are batched or split
    w, h = 3840, 2160
    fs, _ = o.camera((0, 25, 70), (0, 18, 0), w, h, use_scene=False, frames=1, samples=1)
    imgs, counts = [], []
    for batch, calls in [(3, [3]), (2, [3]), (0, [1, 2])]:
        wf.set_option(5, batch)
        wf.allocate(w, h)
        wf.reset_stats()
        f = fs.copy()
        for n in calls:
            wf.invoke(f, spp=n, max_bounces=8)
            f["frames"] += n
            f["samples"] += n
        imgs.append(wf.read_image())
        st = wf.stats()
        counts.append((st["paths"], st["extend_rays"], st["shadow_rays"]))
        assert st["stack_overflows"] == 0
    assert counts[0][0] == 3 * w * h and counts[0] == counts[1] == counts[2]
    assert np.isfinite(imgs[0]).all() and (imgs[0] >= 0).all()
    assert np.array_equal(bits(imgs[1]), bits(imgs[0])) and np.array_equal(bits(imgs[2]), bits(imgs[0]))
    wf.close()


def test_config3_soup_render_against_the_oracle(pa, ob):
    """The 1M-triangle soup rendered, not just probed: 256x144, 2 spp, 4 bounces, next-event estimation — a third of that
    frame is the connect stage.  Image bit-identical to the oracle's; ray, step and triangle-test counters equal."""
    o = ob.Scene("")
    o.soup(1000000)
    o.build()
    mats, tris, nodes = o.arrays()
    w, h, spp = 256, 144, 2
    wf = pa.Wavefront(nee=True, rng_mode=1, traversal_stats=True)
    wf.upload_scene(mats, tris, nodes)
    wf.allocate(w, h)
    fs, _ = o.camera((0, 0, 120), (0, 0, 0), w, h, use_scene=False, frames=1, samples=1)
    wf.invoke(fs, spp=spp, max_bounces=4)
    img = wf.read_image()
    st = wf.stats()
    osum, ost = o.render(ob.render_state(w, h), fs, ob.path_config(4, 1, 1), spp, threads=8)
    assert st["stack_overflows"] == 0
    assert ost["shadow_rays"] > spp * w * h // 2 and ost["extend_rays"] > 3 * spp * w * h // 2
    for k in ["extend_rays", "shadow_rays", "extend_steps", "extend_tris", "shadow_steps", "shadow_tris"]:
        assert st[k] == ost[k], (k, st[k], ost[k])
    assert np.array_equal(bits(img), bits(osum / np.float32(spp)))
    # at the configuration's full size (1920x1080; 6 of its 64 spp): the image is the same bits however the samples are batched
    # or split over calls, the ray counts are equal, nothing overflows
    w, h = 1920, 1080
    fs, _ = o.camera((0, 0, 120), (0, 0, 0), w, h, use_scene=False, frames=1, samples=1)
    imgs, counts = [], []
    for batch, calls in [(6, [6]), (4, [6]), (0, [2, 4])]:
        wf.set_option(5, batch)
        wf.allocate(w, h)
        wf.reset_stats()
        f = fs.copy()
        for n in calls:
            wf.invoke(f, spp=n, max_bounces=4)
            f["frames"] += n
            f["samples"] += n
        imgs.append(wf.read_image())
        st = wf.stats()
        counts.append((st["paths"], st["extend_rays"], st["shadow_rays"], st["extend_steps"], st["shadow_steps"]))
        assert st["stack_overflows"] == 0
    assert counts[0][0] == 6 * w * h and counts[0] == counts[1] == counts[2]
    assert np.isfinite(imgs[0]).all() and (imgs[0] >= 0).all()
    assert np.array_equal(bits(imgs[1]), bits(imgs[0])) and np.array_equal(bits(imgs[2]), bits(imgs[0]))
    wf.close()


# ---- whole frames at the configurations' full sizes against the oracle (not only against themselves)
def _full_frame(pa, ob, scene, view, w, h, spp, bounces, stats=False, threads=16):
    """Renders spp samples of the full-size frame on the GPU and with the oracle over the same arrays; returns both images and
    both counter sets.  The oracle walks arrays the multi-threaded host mirror built (tied to the literal build elsewhere)."""
    mats, tris, nodes = scene.arrays()
    o = ob.Scene("")
    o.set_built(mats, tris, nodes)
    wf = pa.Wavefront(nee=True, rng_mode=1, traversal_stats=stats)
    wf.upload_scene(mats, tris, nodes)
    wf.allocate(w, h)
    fs, _ = o.camera(view[0], view[1], w, h, use_scene=False, frames=1, samples=0)
    wf.invoke(fs, spp=spp, max_bounces=bounces)
    img = wf.read_image()
    st = wf.stats()
    wf.close()
    osum, ost = o.render(ob.render_state(w, h), fs, ob.path_config(bounces, 1, 1), spp, threads=threads)
    return img, st, osum / np.float32(spp), ost


def test_headline_frame_full_size_against_the_oracle(pa, ob):
    """Configuration 2 as bench.py times it — bundled scene, 1920x1080, 4 bounces, next-event estimation — the first 4 of its 16
    samples per pixel at FULL size: every pixel the same bits as the oracle's, ray counts equal (8.3M paths, one batch)."""
    from pathtracer_amd.host import Scene
    s = Scene()
    s.initialize_scene(11)
    s.build(8)
    img, st, oimg, ost = _full_frame(pa, ob, s, ((0, 2, 15), (0, 3, 0)), 1920, 1080, 4, 4)
    assert st["stack_overflows"] == 0 and st["paths"] == 4 * 1920 * 1080
    assert st["extend_rays"] == ost["extend_rays"] and st["shadow_rays"] == ost["shadow_rays"]
    assert np.array_equal(bits(img), bits(oimg))


def test_config3_full_size_frame_against_the_oracle(pa, ob):
    """Configuration 3 at its full 1920x1080, 1 sample per pixel, 4 bounces, next-event estimation: the paths a small frame does
    not take — camera rays queued pixel group by pixel group, the quantised node records, 2M-path launches — against the oracle:
    image bits, ray counts, and the step / triangle-test counters of both traversal stages."""
    from pathtracer_amd.host import Scene
    s = Scene("")
    s.generate_soup(1000000)
    s.build(8)
    img, st, oimg, ost = _full_frame(pa, ob, s, ((0, 0, 120), (0, 0, 0)), 1920, 1080, 1, 4, stats=True)
    assert st["stack_overflows"] == 0
    for k in ["extend_rays", "shadow_rays", "extend_steps", "extend_tris", "shadow_steps", "shadow_tris"]:
        assert st[k] == ost[k], (k, st[k], ost[k])
    assert np.array_equal(bits(img), bits(oimg))


def test_config4_full_size_frame_against_the_oracle(pa, ob):
    """Configuration 4 (259,600 triangles) at its full 3840x2160, 1 sample per pixel, 4 bounces, next-event estimation: image bits
    and ray counts equal the oracle's over all 8.3M pixels."""
    from pathtracer_amd.host import Scene
    s = Scene()
    s.initialize_scene_extended(11, 19)
    s.build(8)
    img, st, oimg, ost = _full_frame(pa, ob, s, ((0, 2, 15), (0, 3, 0)), 3840, 2160, 1, 4)
    assert st["stack_overflows"] == 0 and st["paths"] == 3840 * 2160
    assert st["extend_rays"] == ost["extend_rays"] and st["shadow_rays"] == ost["shadow_rays"]
    assert np.array_equal(bits(img), bits(oimg))


def test_instanced_walk_against_the_flattened_scene(pa, ob):
    """pt_build_scene_instanced + PT_OPT_INSTANCED (no reference counterpart: the reference flattens): the scene as a tree over
    instances and one tree per mesh, triangles still tested in world space on vertices re-derived bit for bit.  Contract of the mode
    (stated in pt_hip.h): hit distances, barycentrics and occlusion flags are the flattened walk's; triangle ids may differ only at
    exact distance ties (a different tie rule) or where the flattened walk, like the reference, drops a hit whose leaf box the ray
    misses by rounding — counted here and bounded at 1 in 10,000; radiance relRMSE against the flattened image <= 1e-3 (measured
    0: the images are the same bits).  Two scenes: 1,000 bunnies of configuration 5's generator (4.97M triangles) and the bundled
    scene, whose floor and light are runs of loose triangles and whose objects are scaled about the world origin."""
    from pathtracer_amd.host import Scene
    for name in ("instances", "bundled"):
        s = Scene()
        if name == "instances":
            s.generate_instances(1000, 11)
            view, lo, hi = ((0, 25, 70), (0, 18, 0)), (-12, 0, -12), (12, 24, 12)
        else:
            s.initialize_scene(11)
            view, lo, hi = ((0, 2, 15), (0, 3, 0)), (-8, 0, -8), (8, 8, 8)
        mats, refs = s.references()
        mesh_tris, mesh_first, mesh_count, inst = s.instancing()
        assert sum(int(mesh_count[i["mesh"]]) for i in inst) == refs.size
        wf = pa.Wavefront(nee=True, rng_mode=1)
        wf.build_scene_instanced(mats, refs, mesh_tris, mesh_first, mesh_count, inst)
        rng = np.random.default_rng(17)
        first = incoherent_rays(rng, 600000, lo, hi)
        h = wf.trace_closest(first)
        ok = np.nonzero(h["tri"] >= 0)[0][:300000]
        rays = incoherent_rays(rng, len(ok), lo, hi)
        rays[:, 0:3] = first[ok, 0:3] + first[ok, 3:6] * h["t"][ok, None]           # bounce rays leaving surfaces
        rays[::50, 0:3] *= np.float32(40.0)                                           # some from far outside: beyond the instanced walk's reach
        rays[::61, 4] = 0.0                                                           # some that can produce NaN slab distances
        tmax = rng.uniform(1, 60, len(ok)).astype(np.float32)
        w, hgt, spp = 256, 144, 4
        got = {}
        for mode in (0, 1):
            wf.set_option(10, mode)
            g = wf.trace_closest(rays)
            occ = wf.trace_occluded(rays, tmax)
            wf.allocate(w, hgt)
            from pathtracer_amd.host import Camera
            fs = Camera(w, hgt).frame_state(view[0], view[1], frames=1)
            wf.reset_stats()
            wf.invoke(fs, spp=spp, max_bounces=6)
            img = wf.read_image()
            st = wf.stats()
            assert st["stack_overflows"] == 0
            got[mode] = (g, occ, img, st)
        (f, focc, fimg, fst), (i, iocc, iimg, ist) = got[0], got[1]
        assert (f["tri"] >= 0).mean() > 0.3
        ids = int((f["tri"] != i["tri"]).sum())
        rel = float(np.sqrt(((fimg[..., :3] - iimg[..., :3]) ** 2).mean()) / np.sqrt((fimg[..., :3] ** 2).mean()))
        pixels = int((bits(fimg) != bits(iimg)).any(axis=-1).sum())
        print(f"{name}: instanced against flattened: {ids} of {len(rays)} ids differ, {pixels} of {w * hgt} pixels, relRMSE {rel:.1e}")
        assert np.array_equal(bits(f["t"]), bits(i["t"])), "a hit distance differs"
        differs = np.nonzero(f["tri"] != i["tri"])[0]
        assert np.array_equal(bits(f["uv"])[f["tri"] == i["tri"]], bits(i["uv"])[f["tri"] == i["tri"]])
        assert ids <= len(rays) // 10000
        assert np.array_equal(focc, iocc)
        assert (fst["extend_rays"], fst["shadow_rays"]) == (ist["extend_rays"], ist["shadow_rays"]) or rel <= 1e-3
        assert rel <= 1e-3
        # what the call refuses: instances that do not tile the list, and a matrix that does not reproduce the triangles
        bad = inst.copy()
        bad["first_triangle"][-1] += 1
        with pytest.raises(pa.PtError) as e:
            wf.build_scene_instanced(mats, refs, mesh_tris, mesh_first, mesh_count, bad)
        assert e.value.code == -1 and "tile" in str(e.value)
        bad = inst.copy()
        bad["transform"][len(inst) // 2][12] += np.float32(0.25)
        with pytest.raises(pa.PtError) as e:
            wf.build_scene_instanced(mats, refs, mesh_tris, mesh_first, mesh_count, bad)
        assert e.value.code == -1 and "bit for bit" in str(e.value)
        # the scene installed before is still there, still instanced
        wf.set_option(10, 1)
        again = wf.trace_closest(rays[:4096])
        assert np.array_equal(again["tri"], i["tri"][:4096])
        wf.close()
