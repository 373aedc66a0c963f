"""ctypes binding of liboracle.so — CPU ORACLE, TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may
import this module.  The product package never does.
"""
import ctypes as C
import os
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle.so")

NODE = np.dtype([("min", "<f4", 3), ("max", "<f4", 3), ("left_first", "<i4"), ("count", "<i4")])
VERTEX = np.dtype([("position", "<f4", 3), ("pad0", "<f4"), ("normal", "<f4", 3), ("pad1", "<f4")])
TRIANGLE = np.dtype([("v1", VERTEX), ("v2", VERTEX), ("v3", VERTEX), ("material", "<i4"), ("pad", "<i4", 3)])
MATERIAL = np.dtype([("color", "<f4", 4), ("emissive", "<f4", 4), ("type", "<i4"), ("index", "<f4"), ("roughness", "<f4"), ("pad", "<i4")])
CAMERA = np.dtype([("p1", "<f4", 3), ("_0", "<f4"), ("p2", "<f4", 3), ("_1", "<f4"), ("p3", "<f4", 3), ("_2", "<f4"),
                   ("position", "<f4", 3), ("_3", "<f4"), ("right", "<f4", 3), ("_4", "<f4"), ("up", "<f4", 3), ("_5", "<f4")])
RENDER_STATE = np.dtype([("width", "<i4"), ("height", "<i4"), ("pad", "<i4", 2), ("sky_color", "<f4", 4)])
FRAME_STATE = np.dtype([("frames", "<i4"), ("samples", "<i4"), ("focal_distance", "<f4"), ("pad", "<i4"), ("camera", CAMERA)])
assert NODE.itemsize == 32 and TRIANGLE.itemsize == 112 and MATERIAL.itemsize == 48
assert RENDER_STATE.itemsize == 32 and FRAME_STATE.itemsize == 112

SKY = (142.0 / 255.0, 178.0 / 255.0, 237.0 / 255.0, 1.0)   # Game.cs:13


def build():
    """Compile liboracle.so (g++, a few seconds)."""
    subprocess.run(["make", "-C", HERE, "-j8"], check=True, stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = C.CDLL(LIB)
        _lib.orc_last_error.restype = C.c_char_p
        _lib.orc_scene_create.restype = C.c_void_p
        _lib.orc_scene_create.argtypes = [C.c_char_p]
        _lib.orc_scene_destroy.argtypes = [C.c_void_p]
    return _lib


def _chk(rc):
    if rc != 0:
        raise RuntimeError("oracle: " + lib().orc_last_error().decode())


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def random_next(seed, n):
    out = np.zeros(n, np.int32)
    _chk(lib().orc_random_next(C.c_int32(seed), n, _p(out)))
    return out


def random_doubles(seed, n):
    out = np.zeros(n, np.float64)
    _chk(lib().orc_random_doubles(C.c_int32(seed), n, _p(out)))
    return out


def det_math(kind, x):
    x = np.ascontiguousarray(x, np.float32)
    out = np.zeros_like(x)
    _chk(lib().orc_det_math({"sin": 0, "cos": 1, "exp": 2}[kind], _p(x), x.size, _p(out)))
    return out


def write_mesh_blob(obj_dir, names, out_path):
    _chk(lib().orc_write_mesh_blob(obj_dir.encode(), ",".join(names).encode(), out_path.encode()))


def render_state(width, height, sky=SKY):
    rs = np.zeros(1, RENDER_STATE)
    rs["width"], rs["height"], rs["sky_color"] = width, height, sky
    return rs


LITERAL_GLSL_EXTEND, LITERAL_PACKED_NORMAL, LITERAL_RUNNING_MEAN = 1, 2, 4


def path_config(max_bounces=10, nee=0, rng_mode=0, tile_rank=0, tile_world=1, literal=0):
    """literal: reference GLSL behaviours the product path departs from (LITERAL_* flags), for the tests that measure the gap."""
    return np.array([max_bounces, nee, rng_mode, tile_rank, tile_world, literal], np.int32)


def second_statement(kind, rows):
    """Two restatements of one helper on the same inputs (rows of 8 floats): see orc_second_statement in capi.cpp."""
    rows = np.ascontiguousarray(rows, np.float32).reshape(-1, 8)
    a, b = np.zeros((rows.shape[0], 3), np.float32), np.zeros((rows.shape[0], 3), np.float32)
    _chk(lib().orc_second_statement({"schlick": 0, "diffuse_direction": 1, "packed_normal": 2, "diffuse_throughput": 3}[kind], _p(rows), rows.shape[0], _p(a), _p(b)))
    return a, b


class Scene:
    """Scene + Accelerator of the oracle (Scene.cs / Accelerator.cs restated)."""

    def __init__(self, mesh_source=""):
        self.h = C.c_void_p(lib().orc_scene_create(mesh_source.encode()))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_scene_destroy(self.h)
            self.h = None

    def clear(self):
        _chk(lib().orc_scene_clear(self.h))

    def initialize_scene(self, seed=11):
        _chk(lib().orc_scene_initialize(self.h, seed))

    def soup(self, n, seed=0x9E3779B9):
        _chk(lib().orc_scene_soup(self.h, n, C.c_uint32(seed)))

    def add_mesh(self, name, normalized=False, matrix=None, material=None):
        m = None if matrix is None else np.ascontiguousarray(matrix, np.float32)
        _chk(lib().orc_scene_add_mesh(self.h, name.encode(), int(normalized), _p(m), _p(material)))

    def floor(self, a, b):
        _chk(lib().orc_scene_floor(self.h, _p(a), _p(b)))

    def add_triangles(self, tris):
        tris = np.ascontiguousarray(tris, TRIANGLE)
        _chk(lib().orc_scene_add_triangles(self.h, _p(tris), tris.size))

    def set_materials(self, mats):
        mats = np.ascontiguousarray(mats, MATERIAL)
        _chk(lib().orc_scene_set_materials(self.h, _p(mats), mats.size))

    def build(self):
        _chk(lib().orc_scene_build(self.h))

    def set_quantized(self, mode):
        """Node records the canonical walks restate: 0 exact, 1 quantised, 2 the GPU library's rule (quantised beyond 32 MiB of records)."""
        _chk(lib().orc_scene_set_quantized(self.h, int(mode)))

    def set_built(self, mats, tris, nodes):
        """Install arrays another builder produced (built order) without building: for trees beyond the literal build's reach."""
        mats, tris, nodes = np.ascontiguousarray(mats, MATERIAL), np.ascontiguousarray(tris, TRIANGLE), np.ascontiguousarray(nodes, NODE)
        _chk(lib().orc_scene_set_built(self.h, _p(mats), mats.size, _p(tris), tris.size, _p(nodes), nodes.size))

    def counts(self):
        out = np.zeros(6, np.int32)
        _chk(lib().orc_scene_counts(self.h, _p(out)))
        return dict(zip(["materials", "triangles", "nodes", "references", "max_depth", "lights"], out.tolist()))

    def arrays(self, references=False):
        c = self.counts()
        mats = np.zeros(c["materials"], MATERIAL)
        tris = np.zeros(c["triangles"], TRIANGLE)
        nodes = np.zeros(c["nodes"], NODE)
        refs = np.zeros(c["references"], TRIANGLE) if references else None
        _chk(lib().orc_scene_copy(self.h, _p(mats), _p(tris), _p(nodes), _p(refs)))
        return (mats, tris, nodes, refs) if references else (mats, tris, nodes)

    def lights(self):
        n = self.counts()["lights"]
        tris, cdf, total = np.zeros(n, np.int32), np.zeros(n, np.float32), C.c_float(0)
        _chk(lib().orc_scene_lights(self.h, _p(tris), _p(cdf), C.byref(total)))
        return tris, cdf, total.value

    def intersect(self, rays, mode="literal", threads=1, stats=False):
        """rays: (n,6) float32 origin+direction. Returns dict(tri, t, uv, point[, stats])."""
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        n = rays.shape[0]
        tri, t = np.zeros(n, np.int32), np.zeros(n, np.float32)
        uv, point = np.zeros((n, 2), np.float32), np.zeros((n, 3), np.float32)
        st = np.zeros(3, np.uint64)
        _chk(lib().orc_intersect(self.h, {"literal": 0, "canonical": 1, "brute": 2}[mode], _p(rays), n,
                                 _p(tri), _p(t), _p(uv), _p(point), _p(st), threads))
        r = dict(tri=tri, t=t, uv=uv, point=point)
        if stats:
            r["stats"] = dict(steps=int(st[0]), tris=int(st[1]), rays=int(st[2]))
        return r

    def intersect_glsl(self, rays, threads=1):
        """The reference's GLSL closest-hit walk and triangle test (extend.glsl:171-299): dict(tri, t, normal, stack_peak, candidate_peak)."""
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        n = rays.shape[0]
        tri, t, nrm, peaks = np.zeros(n, np.int32), np.zeros(n, np.float32), np.zeros((n, 3), np.float32), np.zeros(2, np.int32)
        _chk(lib().orc_intersect_glsl(self.h, _p(rays), n, _p(tri), _p(t), _p(nrm), _p(peaks), threads))
        return dict(tri=tri, t=t, normal=nrm, stack_peak=int(peaks[0]), candidate_peak=int(peaks[1]))

    def occluded(self, rays, tmax, threads=1):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        tmax = np.ascontiguousarray(tmax, np.float32)
        out = np.zeros(rays.shape[0], np.uint8)
        _chk(lib().orc_occluded(self.h, _p(rays), _p(tmax), rays.shape[0], _p(out), threads))
        return out

    def camera(self, position, target, width, height, use_scene=True, frames=0, samples=0):
        """Camera.Set(position, target) -> (FrameState record, focal)."""
        fs = np.zeros(1, FRAME_STATE)
        fs["frames"], fs["samples"] = frames, samples
        pos, tgt = np.asarray(position, np.float32), np.asarray(target, np.float32)
        focal = C.c_float(0)
        _chk(lib().orc_camera(self.h if use_scene else None, _p(pos), _p(tgt), width, height, _p(fs), C.byref(focal)))
        return fs, focal.value

    def render(self, rs, fs, cfg, spp, traversal="canonical", threads=8, max_groups=0, sum_buffer=None, group_stride=1):
        """Adds spp samples into a (H,W,4) sum; returns (sum, stats dict)."""
        w, h = int(rs["width"][0]), int(rs["height"][0])
        s = np.zeros((h, w, 4), np.float32) if sum_buffer is None else sum_buffer
        st = np.zeros(8, np.uint64)
        _chk(lib().orc_render(self.h, _p(rs), _p(fs), _p(cfg), spp, _p(s), _p(st),
                              {"literal": 0, "canonical": 1}[traversal], threads, max_groups, group_stride))
        names = ["extend_rays", "shadow_rays", "extend_steps", "extend_tris", "shadow_steps", "shadow_tris"]
        return s, {k: int(v) for k, v in zip(names, st[:6])}


def generate(rs, fs, cfg, n):
    o_d = np.zeros((n, 6), np.float32)
    pixel = np.zeros(n, np.uint32)
    _chk(lib().orc_generate(_p(rs), _p(fs), _p(cfg), n, _p(o_d), _p(pixel)))
    return o_d, pixel
