"""ctypes loader for the two native libraries of the package.

libpt_hip.so   the HIP kernels behind the C ABI of include/pt_hip.h
libpt_host.so  the C++ host mirror (Accelerator, Scene, Camera) of include/pt_host.h

There is no fallback: a missing library is an error.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))


class NativeLibraryMissing(RuntimeError):
    pass


class pt_config(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("device", C.c_int32), ("nee", C.c_int32), ("rng_mode", C.c_int32),
                ("tile_rank", C.c_int32), ("tile_world", C.c_int32), ("samples_per_batch", C.c_int32),
                ("traversal_stats", C.c_int32), ("stage_timing", C.c_int32), ("device_count", C.c_int32), ("group_flags", C.c_int32),
                ("reserved", C.c_int32 * 5)]


class pt_stats(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("reserved0", C.c_uint32),
                ("paths", C.c_uint64), ("extend_rays", C.c_uint64), ("shadow_rays", C.c_uint64),
                ("extend_steps", C.c_uint64), ("extend_tris", C.c_uint64), ("shadow_steps", C.c_uint64),
                ("shadow_tris", C.c_uint64), ("stack_overflows", C.c_uint64),
                ("extend_launches", C.c_uint32), ("shade_launches", C.c_uint32), ("connect_launches", C.c_uint32),
                ("generate_launches", C.c_uint32),
                ("ms_generate", C.c_float), ("ms_extend", C.c_float), ("ms_shade", C.c_float), ("ms_connect", C.c_float),
                ("ms_render", C.c_float), ("ms_extend_primary", C.c_float), ("reserved", C.c_float * 2),
                ("sched_interior", C.c_uint64), ("sched_leaf", C.c_uint64), ("sched_refill", C.c_uint64),
                ("special_rays", C.c_uint64), ("extend_unique_steps", C.c_uint64), ("extend_unique_tris", C.c_uint64),
                ("shadow_occluded", C.c_uint64), ("device_allocations", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_ if n not in ("reserved", "reserved0", "struct_size")}


PT_OPT_NEE, PT_OPT_RNG_MODE, PT_OPT_TRAVERSAL_STATS, PT_OPT_STAGE_TIMING, PT_OPT_SAMPLES_PER_BATCH, PT_OPT_TRAVERSE_BLOCKS_PER_CU = 1, 2, 3, 4, 5, 6

# every symbol include/pt_hip.h declares
PT_HIP_SYMBOLS = ["pt_default_config", "pt_create", "pt_destroy", "pt_last_error", "pt_upload_scene", "pt_resize",
                  "pt_set_render_state", "pt_reset_accumulation", "pt_render", "pt_synchronize", "pt_read_image",
                  "pt_trace_closest", "pt_trace_occluded", "pt_generate_rays", "pt_get_stats", "pt_reset_stats",
                  "pt_set_option", "pt_bench_extend", "pt_accum_device", "pt_assemble_tiles", "pt_stream", "pt_build_bvh", "pt_build_scene",
                  "pt_trace_closest_noalloc", "pt_build_scene_instanced"]

_hip = None
_host = None


def _load(name):
    path = os.path.join(HERE, name)
    if not os.path.exists(path):
        raise NativeLibraryMissing(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    return C.CDLL(path)


def hip():
    """libpt_hip.so with argument types set."""
    global _hip
    if _hip is None:
        # PyTorch bundles its own libamdhip64.so.7; two HIP runtimes in one process cannot both
        # own the GPU.  Loading torch first makes libpt_hip.so bind to the runtime torch uses
        # (same soname), so tensors, streams and RCCL buffers interoperate with the context.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        # PT_HIP_LIB selects another build of the same library (kernel tuning experiments)
        lib = _load(os.environ.get("PT_HIP_LIB", "libpt_hip.so"))
        vp, i32, u32 = C.c_void_p, C.c_int32, C.c_uint32
        lib.pt_last_error.restype = C.c_char_p
        lib.pt_last_error.argtypes = [vp]
        lib.pt_default_config.argtypes = [C.POINTER(pt_config)]
        lib.pt_create.argtypes = [C.POINTER(pt_config), C.POINTER(vp)]
        lib.pt_destroy.argtypes = [vp]
        lib.pt_upload_scene.argtypes = [vp, vp, i32, vp, i32, vp, i32]
        lib.pt_resize.argtypes = [vp, i32, i32]
        lib.pt_set_render_state.argtypes = [vp, vp]
        lib.pt_reset_accumulation.argtypes = [vp]
        lib.pt_render.argtypes = [vp, vp, i32, i32]
        lib.pt_synchronize.argtypes = [vp]
        lib.pt_read_image.argtypes = [vp, vp]
        lib.pt_trace_closest.argtypes = [vp, vp, i32, vp, vp, vp]
        lib.pt_trace_closest_noalloc.argtypes = [vp, vp, i32, vp, vp, vp]
        lib.pt_trace_occluded.argtypes = [vp, vp, vp, i32, vp]
        lib.pt_generate_rays.argtypes = [vp, vp, i32, vp, vp]
        lib.pt_bench_extend.argtypes = [vp, vp, i32, i32, C.POINTER(C.c_float), vp, vp]
        lib.pt_build_bvh.argtypes = [vp, vp, i32, vp, vp, i32, C.POINTER(i32), C.POINTER(C.c_float)]
        lib.pt_build_scene.argtypes = [vp, vp, i32, vp, i32, vp, vp, i32, C.POINTER(i32), C.POINTER(C.c_float)]
        lib.pt_build_scene_instanced.argtypes = [vp, vp, i32, vp, i32, vp, vp, vp, i32, vp, i32, C.POINTER(C.c_float)]
        lib.pt_get_stats.argtypes = [vp, C.POINTER(pt_stats)]
        lib.pt_reset_stats.argtypes = [vp]
        lib.pt_set_option.argtypes = [vp, i32, i32]
        lib.pt_accum_device.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_uint64), C.POINTER(u32)]
        lib.pt_assemble_tiles.argtypes = [vp, vp, i32, u32, vp]
        lib.pt_stream.argtypes = [vp, C.POINTER(vp)]
        _hip = lib
    return _hip


def host():
    """libpt_host.so (C++ host mirror)."""
    global _host
    if _host is None:
        _host = _load("libpt_host.so")
    return _host
