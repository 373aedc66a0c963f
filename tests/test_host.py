"""Host-side logic without a GPU: the C++ host mirror against the oracle, the C ABI's exported symbols,
and the multi-rank tile path over gloo (world size 2)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import MESHES, ROOT, bundled_view

PKG = os.path.join(ROOT, "pathtracer_amd")


@pytest.fixture(scope="module")
def host():
    from pathtracer_amd import host as h
    return h


def same(a, b):
    return all(x.tobytes() == y.tobytes() for x, y in zip(a, b))


def test_host_bundled_scene_matches_oracle(host, bundled):
    for threads in (1, 8):
        s = host.Scene()
        s.initialize_scene(11)
        s.build(threads)
        assert same(s.arrays(), bundled.arrays())
        assert s.counts()["depth"] == bundled.counts()["max_depth"]


def test_host_other_seeds_and_rebuild(host, ob):
    for seed in (1, 2024):
        s, o = host.Scene(), ob.Scene(MESHES)
        s.initialize_scene(seed)
        o.initialize_scene(seed)
        s.build(4)
        o.build()
        assert same(s.arrays(), o.arrays())
    # Scene.Clear then a new scene (key P, Game.cs:169-180)
    s.clear()
    o.clear()
    s.add_mesh("cube", True)
    o.add_mesh("cube", True)
    s.build()
    o.build()
    assert same(s.arrays(), o.arrays())


def test_host_parallel_build_is_identical_on_soups(host, ob):
    o = ob.Scene("")
    o.soup(120000, seed=99)
    o.build()
    for threads in (1, 3, 8):
        s = host.Scene("")
        s.generate_soup(120000, seed=99)
        s.build(threads)
        assert same(s.arrays(), o.arrays()), threads


def test_host_meshes_materials_floor(host, ob):
    from pathtracer_amd import material, METAL
    a, b = material((0.7, 0.7, 0.9, 0), type=METAL, roughness=0.1), material((0.7, 0.7, 0.9, 0), type=METAL, roughness=0.16)
    rot = np.eye(4, dtype=np.float32)
    rot[3, :3] = (1, 2, 3)
    s, o = host.Scene(), ob.Scene(MESHES)
    for sc, fl in ((s, s.generate_tiled_floor), (o, o.floor)):
        sc.add_mesh("cornell")
        sc.add_mesh("torus", True, rot, a)
        fl(a, b)
        sc.add_mesh("sphere", True, rot * np.float32(0.5), b)
    s.build(2)
    o.build()
    assert same(s.arrays(), o.arrays())
    assert s.get_material_index(a) == s.get_material_index(a.copy())
    n = s.counts()["materials"]
    assert s.get_material_index(material((0.1, 0.2, 0.3, 0))) == n   # a new one is appended


def test_host_synthetic_large_scene_generators(host):
    """The stand-ins for the larger benchmark scenes (Sponza- and San-Miguel-sized): bundled recipe + extra bunnies, and instanced bunnies.  Their BVHs
    obey the builder's invariants and are independent of the thread count."""
    s = host.Scene()
    s.initialize_scene_extended(11, 19)
    s.build(4)
    c = s.counts()
    assert c["triangles"] == 259600            # 165,208 + 19 x 4,968
    mats, tris, nodes = s.arrays()
    base = host.Scene()
    base.initialize_scene(11)
    base.build(1)
    bt = base.arrays()[1]
    # the same random stream: the first 60 objects are the bundled scene's (the build reorders triangles, so compare as sets)
    key = lambda t: np.sort(np.ascontiguousarray(t).view(np.uint8).reshape(t.size, -1).view("V112").ravel())
    assert np.isin(key(bt), key(tris)).all()
    for count in (1, 9):
        a, b = host.Scene(), host.Scene()
        a.generate_instances(count, 11)
        b.generate_instances(count, 11)
        a.build(1)
        b.build(4)
        assert same(a.arrays(), b.arrays())
        m, t, n = a.arrays()
        assert t.size == count * 4968 + 2 and (n.size & 1) == 1
        leaves = n[n["count"] > 0]
        order = np.argsort(leaves["left_first"])
        first, cnt = leaves["left_first"][order], leaves["count"][order]
        assert first[0] == 0 and np.array_equal(first[1:], (first + cnt)[:-1]) and first[-1] + cnt[-1] == t.size   # leaves tile [0, n)
        assert (m["emissive"][:, :3].sum(1) > 0).sum() >= 1


def test_scene_cache_file_round_trip(host, tmp_path):
    s = host.Scene()
    s.initialize_scene(7)
    with pytest.raises(host.HostError):
        s.save(str(tmp_path / "unbuilt.bin"))
    s.build(2)
    path = str(tmp_path / "scene.bin")
    s.save(path)
    c = s.counts()
    assert os.path.getsize(path) == 24 + 48 * c["materials"] + 112 * c["triangles"] + 32 * c["nodes"]
    t = host.Scene("")
    t.load(path)
    assert same(s.arrays(), t.arrays()) and t.counts() == c
    t.build(2)                                 # rebuilding from the stored order gives a valid tree over the same triangles
    assert t.counts()["triangles"] == c["triangles"]
    (tmp_path / "bad.bin").write_bytes(open(path, "rb").read()[:-5])
    with pytest.raises(host.HostError):
        t.load(str(tmp_path / "bad.bin"))
    with pytest.raises(host.HostError):
        t.load(str(tmp_path / "missing.bin"))


def test_references_are_the_insertion_order(host):
    """Scene.references: the triangles as AddTriangle collected them, available before any build and unchanged by it;
    the built array is a permutation of them."""
    s = host.Scene()
    s.add_mesh("cube")
    s.add_mesh("light")
    mats, refs = s.references()
    assert refs.size == 14 and mats.size >= 2
    s.build()
    mats2, refs2 = s.references()
    assert refs2.tobytes() == refs.tobytes() and mats2.tobytes() == mats.tobytes()
    _, tris, _ = s.arrays()
    key = lambda a: sorted(a[i].tobytes() for i in range(a.size))
    assert key(tris) == key(refs)


def test_image_writers(host, tmp_path):
    """PFM and EXR are lossless; PPM and PNG hold what the reference's window would show (clamp, 8 bits, row 0 at the bottom,
    Shaders/Quad/fragment.glsl:9-10) or the sRGB-encoded exposure; the PNG is a valid file (CRC, zlib, Adler)."""
    import struct
    import zlib
    rng = np.random.default_rng(5)
    w, h = 37, 300
    img = rng.uniform(-0.2, 1.4, (h, w, 4)).astype(np.float32)
    img[0, 0] = (np.nan, np.inf, -np.inf, 1)
    img[1, 1] = (0.5, 1.0, 0.0, 1)

    def expected(transfer, exposure):
        v = img[::-1, :, :3].astype(np.float32)
        if transfer == "srgb":
            v = v * np.float32(exposure)
            with np.errstate(invalid="ignore"):
                c = np.clip(np.nan_to_num(v, nan=0.0, posinf=1.0, neginf=0.0), 0, 1)
            v = np.where(c <= 0.0031308, 12.92 * c, 1.055 * np.power(c, 1 / 2.4) - 0.055)
        with np.errstate(invalid="ignore"):
            v = np.clip(np.nan_to_num(v, nan=0.0, posinf=1.0, neginf=0.0), 0, 1)
        return v

    def read_png(path):
        b = open(path, "rb").read()
        assert b[:8] == b"\x89PNG\r\n\x1a\n"
        at, chunks = 8, []
        while at < len(b):
            n, = struct.unpack(">I", b[at:at + 4])
            kind, body = b[at + 4:at + 8], b[at + 8:at + 8 + n]
            assert struct.unpack(">I", b[at + 8 + n:at + 12 + n])[0] == zlib.crc32(kind + body)
            chunks.append((kind, body))
            at += 12 + n
        assert [k for k, _ in chunks] == [b"IHDR", b"IDAT", b"IEND"]
        pw, ph, depth, colour, *_ = struct.unpack(">IIBBBBB", chunks[0][1])
        assert (depth, colour) == (8, 2)
        raw = np.frombuffer(zlib.decompress(chunks[1][1]), np.uint8).reshape(ph, 1 + 3 * pw)
        assert not raw[:, 0].any()
        return raw[:, 1:].reshape(ph, pw, 3)

    pfm = tmp_path / "a.pfm"
    host.write_image(pfm, img)
    b = open(pfm, "rb").read()
    head = b"PF\n%d %d\n-1.0\n" % (w, h)
    assert b.startswith(head)
    assert np.array_equal(np.frombuffer(b[len(head):], np.float32).view(np.uint32).reshape(h, w, 3), img[:, :, :3].view(np.uint32))

    # OpenEXR: magic, version, the attributes a reader requires, the line offset table, one uncompressed chunk per scan line
    exr = tmp_path / "a.exr"
    host.write_image(exr, img)
    b = open(exr, "rb").read()
    assert struct.unpack("<II", b[:8]) == (20000630, 2)
    at, attrs = 8, {}
    while b[at] != 0:
        name_end = b.index(b"\0", at)
        type_end = b.index(b"\0", name_end + 1)
        size, = struct.unpack("<i", b[type_end + 1:type_end + 5])
        attrs[b[at:name_end].decode()] = (b[name_end + 1:type_end].decode(), b[type_end + 5:type_end + 5 + size])
        at = type_end + 5 + size
    at += 1
    assert {"channels", "compression", "dataWindow", "displayWindow", "lineOrder", "pixelAspectRatio", "screenWindowCenter", "screenWindowWidth"} <= set(attrs)
    assert attrs["channels"][0] == "chlist" and attrs["channels"][1] == b"".join(c + b"\0" + struct.pack("<iB3xii", 2, 0, 1, 1) for c in (b"B", b"G", b"R")) + b"\0"
    assert attrs["compression"] == ("compression", b"\0") and attrs["lineOrder"] == ("lineOrder", b"\0")
    assert struct.unpack("<4i", attrs["dataWindow"][1]) == (0, 0, w - 1, h - 1) == struct.unpack("<4i", attrs["displayWindow"][1])
    offsets = struct.unpack("<%dQ" % h, b[at:at + 8 * h])
    assert offsets[0] == at + 8 * h and len(b) == offsets[-1] + 8 + 12 * w
    for y in (0, 1, h - 1):
        yy, nbytes = struct.unpack("<ii", b[offsets[y]:offsets[y] + 8])
        assert (yy, nbytes) == (y, 12 * w)
        chans = np.frombuffer(b[offsets[y] + 8:offsets[y] + 8 + 12 * w], np.uint32).reshape(3, w)   # B, G, R planes of the row
        assert np.array_equal(chans[::-1].T, img[h - 1 - y, :, :3].view(np.uint32))

    for transfer, exposure in (("display", 1.0), ("srgb", 0.75)):
        png, ppm = tmp_path / "a.png", tmp_path / "a.ppm"
        host.write_image(png, img, transfer=transfer, exposure=exposure)
        host.write_image(ppm, img, transfer=transfer, exposure=exposure)
        got = read_png(png)
        assert got.shape == (h, w, 3)
        want = expected(transfer, exposure)
        tol = 0 if transfer == "display" else 1        # powf rounding may move a value across a .5 boundary
        assert np.abs(got.astype(np.int32) - np.floor(want * 255 + 0.5).astype(np.int32)).max() <= tol
        assert tuple(got[h - 1, 0]) == (0, 255, 0) and (transfer != "display" or tuple(got[h - 2, 1]) == (128, 255, 0))
        pb = open(ppm, "rb").read()
        phead = b"P6\n%d %d\n255\n" % (w, h)
        assert pb.startswith(phead) and pb[len(phead):] == got.tobytes()

    big = np.full((200, 150, 4), 0.25, np.float32)     # 200 * 451 = 90,200 raw bytes: two stored deflate blocks
    host.write_image(tmp_path / "b.png", big)
    assert (read_png(tmp_path / "b.png") == 64).all()
    with pytest.raises(host.HostError):
        host.write_image(tmp_path / "no_such_dir" / "c.png", big)
    with pytest.raises(ValueError):
        host.write_image(tmp_path / "c.tiff", big)


def test_host_obj_text_loader(host, ob, tmp_path):
    """The OBJ/MTL text path of the host mirror (Mesh.cs:72-329) on a file written here: quads, negative
    indices, missing normals, tabs, a material library."""
    (tmp_path / "m.mtl").write_text("newmtl red\n  Kd 0.8 0.1 0.1\n  Ke 0 0 0\nnewmtl lamp\n\tKa 0.2 0.2 0.2\n  Kd 1 1 1\n  Ke 3 2 1\n")
    (tmp_path / "box.obj").write_text(
        "# test\nmtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0 0 1.5\nv 1 0 1.5\nv 1 1 1.5\nv 0 1 1.5\n"
        "vn 0 0 -1\ng bottom\nusemtl red\nf 1//1 2//1 3//1 4//1\ng top\nusemtl lamp\nf -4 -3 -2 -1\nf 1/1 2/1 6/1 5/1\n"
        "usemtl missing\nf 2\t3 7 6\n")
    s, o = host.Scene(str(tmp_path) + "/"), ob.Scene(str(tmp_path) + "/")
    for sc in (s, o):
        sc.add_mesh("box", True)
        sc.build()
    assert same(s.arrays(), o.arrays())
    mats, tris, _ = s.arrays()
    assert tris.size == 8 and mats.size == 3
    assert mats["emissive"][1].tolist() == [3, 2, 1, 0]


def test_host_camera_matches_oracle(host, bundled):
    for (w, h) in ((512, 512), (1920, 1080)):
        fs = host.Camera(w, h).frame_state((0, 2, 15), (0, 3, 0), frames=5, samples=5)
        assert fs.tobytes() == bundled_view(bundled, w, h, frames=5).tobytes()
    with pytest.raises(host.HostError):
        host.Scene("/nonexistent/").add_mesh("cube")


def test_c_abi_exports_every_declared_symbol():
    """libpt_hip.so / libpt_host.so load without a GPU and export what include/*.h declares."""
    for header, libname, prefix in (("pt_hip.h", "libpt_hip.so", "pt_"), ("pt_host.h", "libpt_host.so", "pth_")):
        text = open(os.path.join(ROOT, "include", header)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        declared = set(re.findall(r"\b(" + prefix + r"[a-z_0-9]+)\s*\(", text))
        assert len(declared) >= 10
        lib = ctypes.CDLL(os.path.join(PKG, libname))
        for name in sorted(declared):
            assert hasattr(lib, name), f"{libname} does not export {name}"
    from pathtracer_amd import _native
    assert set(_native.PT_HIP_SYMBOLS) == set(re.findall(r"\b(pt_[a-z_0-9]+)\s*\(", re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "pt_hip.h")).read(), flags=re.S)))


def test_csharp_shim_matches_the_header():
    """The C# shim cannot be compiled here, so what can drift is checked as text: the files under bindings/csharp are the code
    blocks of INTEGRATION.md; every DllImport names an exported function and takes as many parameters as the C declaration; the
    Config struct has pt_config's fields in pt_config's order."""
    from pathtracer_amd import _native
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    native = open(os.path.join(ROOT, "bindings", "csharp", "Native.cs")).read()
    for name in os.listdir(os.path.join(ROOT, "bindings", "csharp")):
        body = open(os.path.join(ROOT, "bindings", "csharp", name)).read().split("\n", 2)[2]      # below the two header lines
        for line in (l for l in body.splitlines() if l.strip()):
            assert line in doc, f"bindings/csharp/{name} has a line INTEGRATION.md does not: run tools/extract_csharp.py\n{line}"
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "pt_hip.h")).read(), flags=re.S)
    c_params = {m.group(1): len([a for a in m.group(2).split(",") if a.strip() and a.strip() != "void"])
                for m in re.finditer(r"\b(pt_[a-z_0-9]+)\s*\(([^)]*)\)\s*;", header)}
    imports = re.findall(r"\[DllImport\(Lib\)\]\s*public static extern \w+ (pt_[a-z_0-9]+)\(([^)]*)\)", native, flags=re.S)
    assert len(imports) >= 12
    lib = ctypes.CDLL(os.path.join(PKG, "libpt_hip.so"))
    for fn, params in imports:
        assert hasattr(lib, fn), fn
        assert len([a for a in params.split(",") if a.strip()]) == c_params[fn], fn
    struct = re.search(r"public struct Config\s*\{(.*?)\n        \}", native, flags=re.S).group(1)
    struct = re.sub(r"//[^\n]*", "", struct)
    fields = []
    for decl in re.findall(r"public int(\[\])? ([^;]*);", struct):
        fields += [f.strip() for f in decl[1].split(",")]
    snake = lambda f: re.sub(r"(?<!^)([A-Z])", r"_\1", f).lower()
    assert [snake(f) for f in fields] == [n for n, _ in _native.pt_config._fields_]
    assert "SizeConst = 5" in native and ctypes.sizeof(_native.pt_config) == 4 * (len(fields) - 1 + 5)
    # every function the header declares has an import
    assert sorted(c_params) == sorted(_native.PT_HIP_SYMBOLS), "pathtracer_amd/_native.py's symbol list and pt_hip.h disagree"
    assert sorted(fn for fn, _ in imports) == sorted(c_params), sorted(set(c_params) - {fn for fn, _ in imports})
    # the Stats struct has pt_stats' fields, types and order
    stats = re.search(r"public struct Stats\s*\{(.*?)\n        \}", native, flags=re.S).group(1)
    stats = re.sub(r"//[^\n]*", "", stats)
    cs_fields = []
    for ty, names in re.findall(r"public (uint|ulong|float) ([^;]*);", stats):
        cs_fields += [(n.strip(), ty) for n in names.split(",")]
    want = []
    for name, ctype in _native.pt_stats._fields_:
        if name == "reserved":
            want += [("reserved1", "float"), ("reserved2", "float")]
        else:
            want.append((name, {ctypes.c_uint32: "uint", ctypes.c_uint64: "ulong", ctypes.c_float: "float"}[ctype]))
    assert [(snake(n), t) for n, t in cs_fields] == want
    # the option keys
    for key, value in re.findall(r"PT_OPT_([A-Z_]+) = (\d+)", header):
        camel = "Opt" + "".join(w.capitalize() for w in key.split("_"))
        assert re.search(rf"\b{camel} = {value}\b", native), camel


def test_ctypes_structs_match_the_header(tmp_path):
    """pt_config and pt_stats as the Python binding (and, through it, the C# shim) lays them out, against what a C compiler makes of
    include/pt_hip.h: size and the offset of every field."""
    from pathtracer_amd import _native
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "pt_hip.h"', 'int main(void) {']
    for cname, st in (("pt_config", _native.pt_config), ("pt_stats", _native.pt_stats)):
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for name, _ in st._fields_:
            lines.append(f'printf("{cname}.{name} %zu\\n", offsetof({cname}, {name}));')
    lines.append("return 0; }")
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-I" + os.path.join(ROOT, "include"), "-o", str(exe), str(src)], check=True)
    out = dict(l.split() for l in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    for cname, st in (("pt_config", _native.pt_config), ("pt_stats", _native.pt_stats)):
        assert int(out[cname]) == ctypes.sizeof(st)
        for name, _ in st._fields_:
            assert int(out[f"{cname}.{name}"]) == getattr(st, name).offset, (cname, name)


def test_headers_are_plain_c(tmp_path):
    """include/*.h compile as C99 with -pedantic, and the C example builds against both libraries."""
    src = tmp_path / "h.c"
    src.write_text('#include "pt_hip.h"\n#include "pt_host.h"\nint main(void) { pt_config c; return pt_default_config(&c) == PT_OK ? (int)sizeof(pt_stats) * 0 : 1; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert os.path.exists(os.path.join(ROOT, "examples", "render_c")), "run __graft_entry__.build() first"


def test_missing_library_is_an_error(monkeypatch):
    from pathtracer_amd import _native
    monkeypatch.setattr(_native, "_hip", None)
    monkeypatch.setenv("PT_HIP_LIB", "libpt_hip_absent.so")
    with pytest.raises(_native.NativeLibraryMissing):
        _native.hip()
    monkeypatch.setattr(_native, "_hip", None)


def test_tile_partition_helpers():
    from pathtracer_amd import tiles
    own = tiles.tile_owner(40, 24, 3)
    assert own.shape == (24, 40) and set(np.unique(own)) == {0, 1, 2}
    assert (own[:8, :8] == 0).all() and (own[:8, 8:16] == 1).all() and (own[8:16, :8] == (5 % 3)).all()
    total = sum(tiles.local_slots(40, 24, r, 3)[0] for r in range(3))
    assert total == 5 * 3 * 64
    assert len({tiles.local_slots(40, 24, r, 3)[1] for r in range(3)}) == 1


WORKER = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, {root!r})
sys.path.insert(0, os.path.join({root!r}, "tests"))
from oracle import binding as ob
from pathtracer_amd import tiles
from conftest import MESHES

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
w, h, spp = 72, 40, 2
s = ob.Scene(MESHES); s.add_mesh("cornell"); s.build()
fs, _ = s.camera((0, 1, 3.4), (0, 1, 0), w, h, use_scene=True, frames=1, samples=1)
rs = ob.render_state(w, h, sky=(0.1, 0.2, 0.3, 1))
# this rank's tiles only, laid out as the device accumulator is: slot = local group * 64 + lane
own, padded = tiles.local_slots(w, h, rank, world)
img, _ = s.render(rs, fs, ob.path_config(3, 1, 1, tile_rank=rank, tile_world=world), spp, threads=2)
gcx = (w + 7) // 8
local = np.zeros((padded, 4), np.float32)
def compact(x):
    x &= 0x55555555; x = (x ^ (x >> 1)) & 0x33333333; x = (x ^ (x >> 2)) & 0x0f0f0f0f; x = (x ^ (x >> 4)) & 0x00ff00ff; x = (x ^ (x >> 8)) & 0xffff
    return x
for slot in range(own):
    g = (slot >> 6) * world + rank; lid = slot & 63
    x, y = (g % gcx) * 8 + compact(lid), (g // gcx) * 8 + compact(lid >> 1)
    if x < w and y < h:
        local[slot] = img[y, x]
gathered = tiles.gather_accumulators(torch.from_numpy(local), dst=0)
if rank == 0:
    assert gathered.shape == (world, padded, 4)
    out = np.zeros((h, w, 4), np.float32)
    own_map = tiles.tile_owner(w, h, world)
    for y in range(h):
        for x in range(w):
            g = (y // 8) * gcx + (x // 8)
            r, lg = g % world, g // world
            lid = 0
            for b in range(3):
                lid |= (((x % 8) >> b) & 1) << (2 * b) | (((y % 8) >> b) & 1) << (2 * b + 1)
            assert own_map[y, x] == r
            out[y, x] = gathered[r, lg * 64 + lid].numpy()
    whole, _ = s.render(rs, fs, ob.path_config(3, 1, 1), spp, threads=2)
    assert np.array_equal(out.view(np.uint32), whole.view(np.uint32)), "gathered tiles differ from the single-rank image"
    print("GLOO_TILES_OK")
else:
    assert gathered is None
dist.barrier()
dist.destroy_process_group()
"""


def test_two_rank_tile_gather_over_gloo(tmp_path):
    """world_size 2 on CPU: each rank renders its own tiles (oracle), tiles.gather_accumulators moves them to rank 0,
    and the de-interleaved image equals the single-rank image bit for bit."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29613", str(script)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "GLOO_TILES_OK" in r.stdout


def test_isa_cost_file_is_current():
    """pathtracer_amd/isa_cost.json (VALU instructions per traversal iteration, which bench.py prices the roofline with) must be
    what tools/isa_cost.py derives from the kernels' assembly today."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_cost.py"), "--check"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_scene_as_meshes_and_instances():
    """pth_scene_instancing (input of pt_build_scene_instanced): the instances tile the flattened triangle list in order, and
    Vector3.Transform(mesh vertex, instance matrix) — ((x M11 + y M21) + z M31) + M41, Mesh.cs:50-70 — reproduces every flattened
    vertex bit for bit, for the instanced-bunny generator and for the bundled recipe (scale applied after the translation)."""
    from pathtracer_amd.host import Scene
    for make in (lambda s: s.generate_instances(30, 11), lambda s: s.initialize_scene(11)):
        s = Scene()
        make(s)
        _, refs = s.references()
        tris, first, count, inst = s.instancing()
        cursor = 0
        for it in inst:
            m, n = int(it["mesh"]), int(count[it["mesh"]])
            assert int(it["first_triangle"]) == cursor
            cursor += n
            M = it["transform"].reshape(4, 4).astype(np.float32)
            mesh, flat = tris[first[m]:first[m] + n], refs[cursor - n:cursor]
            for v in ("v1", "v2", "v3"):
                p = mesh[v]["position"].astype(np.float32)
                w = np.stack([((p[:, 0] * M[0, a] + p[:, 1] * M[1, a]) + p[:, 2] * M[2, a]) + M[3, a] for a in range(3)], 1).astype(np.float32)
                assert np.array_equal(w.view(np.uint32), flat[v]["position"].view(np.uint32))
        assert cursor == refs.size
