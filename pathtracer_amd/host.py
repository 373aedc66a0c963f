"""Python face of libpt_host.so: the reference's host classes for machines without the C# host.

Scene   PathTracer/Scene.cs + the recipe of PathTracer/Game.cs:66-136 (+ Accelerator.Build)
Camera  PathTracer/Camera.cs (Set / Update / GetFocalDistance / SetUniform)

All arithmetic is in the C++ library; the autofocus ray of Camera.GetFocalDistance is traced on the
GPU through Wavefront.trace_closest (pt_trace_closest), as Scene.Intersect's twin.
"""
import ctypes as C
import os

import numpy as np

from . import _native
from .types import FRAME_STATE, INSTANCE, MATERIAL, NODE, TRIANGLE

ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", "meshes.bin")


class HostError(RuntimeError):
    pass


def _lib():
    lib = _native.host()
    if not getattr(lib, "_pth_typed", False):
        vp, i32 = C.c_void_p, C.c_int32
        lib.pth_last_error.restype = C.c_char_p
        lib.pth_scene_create.restype = vp
        lib.pth_scene_create.argtypes = [C.c_char_p]
        lib.pth_scene_destroy.argtypes = [vp]
        lib.pth_scene_clear.argtypes = [vp]
        lib.pth_scene_add_mesh.argtypes = [vp, C.c_char_p, C.c_int, vp, vp]
        lib.pth_scene_generate_tiled_floor.argtypes = [vp, vp, vp]
        lib.pth_scene_get_material_index.argtypes = [vp, vp, C.POINTER(i32)]
        lib.pth_scene_add_triangles.argtypes = [vp, vp, i32]
        lib.pth_scene_initialize.argtypes = [vp, i32]
        lib.pth_scene_generate_soup.argtypes = [vp, i32, C.c_uint32]
        lib.pth_scene_initialize_extended.argtypes = [vp, i32, i32]
        lib.pth_scene_generate_instances.argtypes = [vp, i32, i32]
        lib.pth_scene_build.argtypes = [vp, i32]
        lib.pth_scene_save.argtypes = [vp, C.c_char_p]
        lib.pth_scene_load.argtypes = [vp, C.c_char_p]
        lib.pth_scene_counts.argtypes = [vp, vp]
        lib.pth_scene_references.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(i32), C.POINTER(i32)]
        lib.pth_scene_arrays.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
        lib.pth_scene_instancing.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), vp, vp, vp, vp]
        lib.pth_camera_basis.argtypes = [vp, vp, vp, vp, vp]
        lib.pth_camera_state.argtypes = [vp, vp, vp, vp, C.c_float, i32, i32, vp, C.POINTER(C.c_float)]
        lib.pth_image_write.argtypes = [C.c_char_p, vp, i32, i32, i32, i32, C.c_float]
        lib._pth_typed = True
    return lib


def _chk(rc):
    if rc != 0:
        raise HostError(_lib().pth_last_error().decode())


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


IMAGE_FORMATS = {"pfm": 0, "ppm": 1, "png": 2, "exr": 3}
TRANSFERS = {"display": 0, "srgb": 1}


def write_image(path, rgba, transfer="display", exposure=1.0, format=None):
    """Write the (H, W, 4) float32 read-out of Wavefront.read_image (row 0 = bottom) as .pfm or .exr (float32, lossless), .ppm or .png.
    transfer="display" is what the reference's window shows (Shaders/Quad/fragment.glsl:9-10: clamp, 8 bits, no gamma);
    "srgb" applies exposure and the sRGB curve."""
    rgba = np.ascontiguousarray(rgba, dtype=np.float32)
    if rgba.ndim != 3 or rgba.shape[2] != 4:
        raise ValueError("rgba must have shape (height, width, 4)")
    fmt = format or os.path.splitext(str(path))[1].lstrip(".").lower()
    if fmt not in IMAGE_FORMATS or transfer not in TRANSFERS:
        raise ValueError("unknown image format %r or transfer %r" % (fmt, transfer))
    _chk(_lib().pth_image_write(os.fsencode(path), _p(rgba), rgba.shape[1], rgba.shape[0], IMAGE_FORMATS[fmt], TRANSFERS[transfer], float(exposure)))


class Scene:
    def __init__(self, mesh_source=ASSETS):
        self._lib = _lib()
        self._h = C.c_void_p(self._lib.pth_scene_create(mesh_source.encode()))

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.pth_scene_destroy(self._h)
            self._h = None

    def clear(self):
        _chk(self._lib.pth_scene_clear(self._h))

    def add_mesh(self, name, normalized=False, matrix=None, material=None):
        m = None if matrix is None else np.ascontiguousarray(matrix, np.float32)
        mat = None if material is None else np.ascontiguousarray(material, MATERIAL)
        _chk(self._lib.pth_scene_add_mesh(self._h, name.encode(), int(normalized), _p(m), _p(mat)))

    def generate_tiled_floor(self, a, b):
        a, b = np.ascontiguousarray(a, MATERIAL), np.ascontiguousarray(b, MATERIAL)
        _chk(self._lib.pth_scene_generate_tiled_floor(self._h, _p(a), _p(b)))

    def get_material_index(self, material):
        mat = np.ascontiguousarray(material, MATERIAL)
        idx = C.c_int32()
        _chk(self._lib.pth_scene_get_material_index(self._h, _p(mat), C.byref(idx)))
        return idx.value

    def add_triangles(self, tris):
        tris = np.ascontiguousarray(tris, TRIANGLE)
        _chk(self._lib.pth_scene_add_triangles(self._h, _p(tris), tris.size))

    def initialize_scene(self, seed=11):
        _chk(self._lib.pth_scene_initialize(self._h, seed))

    def initialize_scene_extended(self, seed=11, extra_bunnies=19):
        _chk(self._lib.pth_scene_initialize_extended(self._h, seed, extra_bunnies))

    def generate_instances(self, count, seed=11):
        _chk(self._lib.pth_scene_generate_instances(self._h, count, seed))

    def generate_soup(self, n_triangles, seed=0x9E3779B9):
        _chk(self._lib.pth_scene_generate_soup(self._h, n_triangles, C.c_uint32(seed)))

    def build(self, threads=1):
        _chk(self._lib.pth_scene_build(self._h, threads))

    def save(self, path):
        """Write the built scene (materials, triangles in built order, nodes) to a cache file."""
        _chk(self._lib.pth_scene_save(self._h, os.fsencode(path)))

    def load(self, path):
        """Replace the scene by a cache file written by save(); no build is needed afterwards."""
        _chk(self._lib.pth_scene_load(self._h, os.fsencode(path)))

    def counts(self):
        out = np.zeros(4, np.int32)
        _chk(self._lib.pth_scene_counts(self._h, _p(out)))
        return dict(zip(["materials", "triangles", "nodes", "depth"], out.tolist()))

    def references(self):
        """Copies of (Material[], Triangle[]) in insertion order, before any build: the input of Wavefront.build_scene."""
        pt, pm, nt, nm = C.c_void_p(), C.c_void_p(), C.c_int32(), C.c_int32()
        _chk(self._lib.pth_scene_references(self._h, C.byref(pt), C.byref(pm), C.byref(nt), C.byref(nm)))

        def view(ptr, n, dt):
            if n == 0:
                return np.zeros(0, dt)
            return np.frombuffer((C.c_char * (n * dt.itemsize)).from_address(ptr.value), dtype=dt, count=n).copy()
        return view(pm, nm.value, MATERIAL), view(pt, nt.value, TRIANGLE)

    def instancing(self):
        """The scene as meshes and instances (see pth_scene_instancing): (mesh_triangles, mesh_first, mesh_count, instances) for
        Wavefront.build_scene_instanced; instances is a structured array (transform 16 floats, mesh, first_triangle)."""
        nm, nt, ni = C.c_int32(), C.c_int32(), C.c_int32()
        _chk(self._lib.pth_scene_instancing(self._h, C.byref(nm), C.byref(nt), C.byref(ni), None, None, None, None))
        tris = np.zeros(nt.value, TRIANGLE)
        first, count = np.zeros(nm.value, np.int32), np.zeros(nm.value, np.int32)
        inst = np.zeros(ni.value, INSTANCE)
        _chk(self._lib.pth_scene_instancing(self._h, None, None, None, _p(tris), _p(first), _p(count), _p(inst)))
        return tris, first, count, inst

    def arrays(self):
        """Copies of (Material[], Triangle[], Node[]) — what Scene.CopyToDevice uploads."""
        c = self.counts()
        pm, pt, pn = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _chk(self._lib.pth_scene_arrays(self._h, C.byref(pm), C.byref(pt), C.byref(pn)))

        def view(ptr, n, dt):
            if n == 0:
                return np.zeros(0, dt)
            buf = (C.c_char * (n * dt.itemsize)).from_address(ptr.value)
            return np.frombuffer(buf, dtype=dt, count=n).copy()
        return view(pm, c["materials"], MATERIAL), view(pt, c["triangles"], TRIANGLE), view(pn, c["nodes"], NODE)

    def copy_to_device(self, wavefront, threads=1):
        """Scene.CopyToDevice, Scene.cs:176-183: build, then upload the three buffers."""
        self.build(threads)
        wavefront.upload_scene(*self.arrays())


class Camera:
    """Camera.cs.  `wavefront` (optional) supplies the autofocus ray; without it the scene is taken to be empty."""

    def __init__(self, width, height, wavefront=None):
        self.width, self.height, self.wavefront = width, height, wavefront
        self.focal = 1.0
        self.camera_state = None

    def set(self, position, target):
        lib = _lib()
        pos, tgt = np.asarray(position, np.float32), np.asarray(target, np.float32)
        f, r, u = np.zeros(3, np.float32), np.zeros(3, np.float32), np.zeros(3, np.float32)
        _chk(lib.pth_camera_basis(_p(pos), _p(tgt), _p(f), _p(r), _p(u)))
        dist = np.float32(np.inf)
        if self.wavefront is not None:
            dist = self.wavefront.trace_closest(np.concatenate([pos, f])[None, :])["t"][0]
        cs = np.zeros(96, np.uint8)
        focal = C.c_float()
        _chk(lib.pth_camera_state(_p(pos), _p(f), _p(r), _p(u), C.c_float(float(dist)), self.width, self.height, _p(cs), C.byref(focal)))
        self.focal = focal.value
        self.camera_state = cs
        return self

    def frame_state(self, position, target, frames=0, samples=0):
        """Camera.Set + SetUniform: a FrameState record ready for Wavefront.invoke."""
        self.set(position, target)
        fs = np.zeros(1, FRAME_STATE)
        fs["frames"], fs["samples"] = frames, samples
        fs.view(np.uint8)[16:112] = self.camera_state
        return fs
