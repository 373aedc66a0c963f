"""Host driver of the wavefront stages — the Python face of libpt_hip.so.

Mirrors PathTracer/Wavefront.cs (Allocate / Invoke) and the parts of PathTracer/Game.cs that
feed it (Resize, Draw's frames/samples bookkeeping).  All compute is in the HIP library; this
module only marshals the host's arrays through the C ABI.
"""
import ctypes as C
import numpy as np

from . import _native
from .types import FRAME_STATE, MATERIAL, NODE, RENDER_STATE, TRIANGLE, render_state


class PtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"pt_hip error {code}: {message}")
        self.code = code


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Wavefront:
    """One rendering context on one GPU (pt_ctx)."""

    def __init__(self, device=0, nee=False, rng_mode=0, tile_rank=0, tile_world=1, samples_per_batch=0,
                 traversal_stats=False, stage_timing=False, device_count=1, group_flags=0):
        self._lib = _native.hip()
        cfg = _native.pt_config()
        self._check(self._lib.pt_default_config(C.byref(cfg)), None)
        cfg.device, cfg.nee, cfg.rng_mode = device, int(nee), rng_mode
        cfg.tile_rank, cfg.tile_world, cfg.samples_per_batch = tile_rank, tile_world, samples_per_batch
        cfg.traversal_stats, cfg.stage_timing = int(traversal_stats), int(stage_timing)
        cfg.device_count, cfg.group_flags = device_count, group_flags
        self.tile_rank, self.tile_world = tile_rank, tile_world
        h = C.c_void_p()
        self._h = None
        self._check(self._lib.pt_create(C.byref(cfg), C.byref(h)), None)
        self._h = h
        self.width = self.height = 0

    def _check(self, rc, h="self"):
        if rc != 0:
            handle = self._h if h == "self" else None
            raise PtError(rc, self._lib.pt_last_error(handle).decode())

    def close(self):
        if self._h is not None:
            self._lib.pt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- scene (Scene.CopyToDevice)
    def upload_scene(self, materials, triangles, nodes):
        materials = np.ascontiguousarray(materials, MATERIAL)
        triangles = np.ascontiguousarray(triangles, TRIANGLE)
        nodes = np.ascontiguousarray(nodes, NODE)
        self._check(self._lib.pt_upload_scene(self._h, _ptr(materials), materials.size, _ptr(triangles), triangles.size,
                                              _ptr(nodes), nodes.size))

    # ---- Game.Resize
    def allocate(self, width, height, sky=None):
        self._check(self._lib.pt_resize(self._h, width, height))
        self.width, self.height = width, height
        rs = render_state(width, height) if sky is None else render_state(width, height, sky)
        self.set_render_state(rs)

    def set_render_state(self, rs):
        rs = np.ascontiguousarray(rs, RENDER_STATE)
        self._check(self._lib.pt_set_render_state(self._h, _ptr(rs)))

    def reset_accumulation(self):
        self._check(self._lib.pt_reset_accumulation(self._h))

    def set_option(self, key, value):
        self._check(self._lib.pt_set_option(self._h, key, int(value)))

    # ---- Game.Draw -> Wavefront.Invoke
    def invoke(self, frame_state, spp=1, max_bounces=10, wait=True):
        fs = np.ascontiguousarray(frame_state, FRAME_STATE)
        self._check(self._lib.pt_render(self._h, _ptr(fs), spp, max_bounces))
        if wait:
            self.synchronize()

    def synchronize(self):
        self._check(self._lib.pt_synchronize(self._h))

    def read_image(self):
        img = np.zeros((self.height, self.width, 4), np.float32)
        self._check(self._lib.pt_read_image(self._h, _ptr(img)))
        return img

    # ---- queries
    def trace_closest(self, rays):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        n = rays.shape[0]
        tri, t, uv = np.zeros(n, np.int32), np.zeros(n, np.float32), np.zeros((n, 2), np.float32)
        self._check(self._lib.pt_trace_closest(self._h, _ptr(rays), n, _ptr(tri), _ptr(t), _ptr(uv)))
        return dict(tri=tri, t=t, uv=uv)

    def trace_closest_noalloc(self, rays):
        """trace_closest through the entry point that allocates no device memory (at most 65,536 rays per call)."""
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        n = rays.shape[0]
        tri, t, uv = np.zeros(n, np.int32), np.zeros(n, np.float32), np.zeros((n, 2), np.float32)
        self._check(self._lib.pt_trace_closest_noalloc(self._h, _ptr(rays), n, _ptr(tri), _ptr(t), _ptr(uv)))
        return dict(tri=tri, t=t, uv=uv)

    def trace_occluded(self, rays, tmax):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        tmax = np.ascontiguousarray(tmax, np.float32)
        out = np.zeros(rays.shape[0], np.uint8)
        self._check(self._lib.pt_trace_occluded(self._h, _ptr(rays), _ptr(tmax), rays.shape[0], _ptr(out)))
        return out

    def generate_rays(self, frame_state, n):
        fs = np.ascontiguousarray(frame_state, FRAME_STATE)
        o_d, pixel = np.zeros((n, 6), np.float32), np.zeros(n, np.uint32)
        self._check(self._lib.pt_generate_rays(self._h, _ptr(fs), n, _ptr(o_d), _ptr(pixel)))
        return o_d, pixel

    def bench_extend(self, rays, iters=5):
        """Mean launch time (ms) of the extend stage alone over `rays`, plus the hits of the last launch."""
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        n = rays.shape[0]
        ms = C.c_float()
        tri, t = np.zeros(n, np.int32), np.zeros(n, np.float32)
        self._check(self._lib.pt_bench_extend(self._h, _ptr(rays), n, iters, C.byref(ms), _ptr(tri), _ptr(t)))
        return ms.value, dict(tri=tri, t=t)

    def build_bvh(self, triangles):
        """Accelerator.Build on the GPU: (Triangle[] reordered, Node[], device milliseconds) for triangles in insertion order."""
        from .types import NODE, TRIANGLE
        tris = np.ascontiguousarray(triangles, TRIANGLE)
        n = tris.size
        cap = max(1, 2 * n - 1)
        out_t, out_n = np.zeros(n, TRIANGLE), np.zeros(cap, NODE)
        nn, ms = C.c_int32(), C.c_float()
        self._check(self._lib.pt_build_bvh(self._h, _ptr(tris), n, _ptr(out_t), _ptr(out_n), cap, C.byref(nn), C.byref(ms)))
        return out_t, out_n[:nn.value].copy(), ms.value

    def build_scene(self, materials, triangles, want_arrays=False):
        """Build the BVH on the GPU and install the scene without a host round trip (pt_build_scene).
        Returns device milliseconds, or (Triangle[], Node[], ms) with want_arrays."""
        from .types import MATERIAL, NODE, TRIANGLE
        mats = np.ascontiguousarray(materials, MATERIAL)
        tris = np.ascontiguousarray(triangles, TRIANGLE)
        n = tris.size
        cap = max(1, 2 * n - 1)
        out_t = np.zeros(n, TRIANGLE) if want_arrays else None
        out_n = np.zeros(cap, NODE) if want_arrays else None
        nn, ms = C.c_int32(), C.c_float()
        self._check(self._lib.pt_build_scene(self._h, _ptr(mats), mats.size, _ptr(tris), n, _ptr(out_t), _ptr(out_n), cap, C.byref(nn), C.byref(ms)))
        return (out_t, out_n[:nn.value].copy(), ms.value) if want_arrays else ms.value

    def build_scene_instanced(self, materials, triangles, mesh_triangles, mesh_first, mesh_count, instances):
        """pt_build_scene plus the scene as instances of meshes (pt_build_scene_instanced); the arguments after `triangles` are
        what host.Scene.instancing() returns.  set_option(10, 1) then makes the kernels walk the instances.  Returns device ms."""
        from .types import INSTANCE, MATERIAL, TRIANGLE
        mats = np.ascontiguousarray(materials, MATERIAL)
        tris = np.ascontiguousarray(triangles, TRIANGLE)
        mtris = np.ascontiguousarray(mesh_triangles, TRIANGLE)
        first, count = np.ascontiguousarray(mesh_first, np.int32), np.ascontiguousarray(mesh_count, np.int32)
        inst = np.ascontiguousarray(instances, INSTANCE)
        ms = C.c_float()
        self._check(self._lib.pt_build_scene_instanced(self._h, _ptr(mats), mats.size, _ptr(tris), tris.size, _ptr(mtris), _ptr(first), _ptr(count), first.size,
                                                       _ptr(inst), inst.size, C.byref(ms)))
        return ms.value

    def stats(self):
        s = _native.pt_stats()
        s.struct_size = C.sizeof(_native.pt_stats)
        self._check(self._lib.pt_get_stats(self._h, C.byref(s)))
        return s.as_dict()

    def reset_stats(self):
        self._check(self._lib.pt_reset_stats(self._h))

    # ---- multi-GPU tiles
    def accum_device(self):
        """(device pointer, padded slot count, samples) of the rank-local float4 accumulator."""
        p, n, s = C.c_void_p(), C.c_uint64(), C.c_uint32()
        self._check(self._lib.pt_accum_device(self._h, C.byref(p), C.byref(n), C.byref(s)))
        return p.value, n.value, s.value

    def stream(self):
        """The hipStream_t (as an integer) the context enqueues on."""
        p = C.c_void_p()
        self._check(self._lib.pt_stream(self._h, C.byref(p)))
        return p.value or 0

    def assemble_tiles(self, gathered_device_ptr, world, samples):
        img = np.zeros((self.height, self.width, 4), np.float32)
        self._check(self._lib.pt_assemble_tiles(self._h, C.c_void_p(gathered_device_ptr), world, samples, _ptr(img)))
        return img
