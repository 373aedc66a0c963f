"""The entry point that allocates no device memory (pt_trace_closest_noalloc) against pt_trace_closest.

Collected last on purpose (file name): it shares one fixed device array per GPU with every other caller in the process.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rays(rng, n):
    rays = np.zeros((n, 6), np.float32)
    rays[:, 0:3] = rng.uniform(-8, 8, (n, 3)).astype(np.float32) + np.array([0, 9, 0], np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    rays[:, 3:6] = d / np.sqrt((d * d).sum(1, dtype=np.float32))[:, None]
    rays[::97, 4] = 0.0     # a few rays on the NaN-rejecting walk
    return rays


def test_noalloc_trace_equals_trace_closest(bundled):
    """Two calls in a row through the C ABI, the second with fewer rays (the array keeps the first call's contents):
    triangle ids, distances and barycentrics bit-identical to pt_trace_closest; sizes beyond the fixed array are refused."""
    import pathtracer_amd as pa
    wf = pa.Wavefront()
    mats, tris, nodes = bundled.arrays()
    wf.upload_scene(mats, tris, nodes)
    rng = np.random.default_rng(23)
    for n in (65536, 1000):
        rays = _rays(rng, n)
        want = wf.trace_closest(rays)
        got = wf.trace_closest_noalloc(rays)
        assert (want["tri"] >= 0).sum() > n // 4
        assert np.array_equal(got["tri"], want["tri"])
        assert np.array_equal(got["t"].view(np.uint32), want["t"].view(np.uint32))
        assert np.array_equal(got["uv"].view(np.uint32), want["uv"].view(np.uint32))
    with pytest.raises(pa.PtError) as e:
        wf.trace_closest_noalloc(_rays(rng, 65537))
    assert e.value.code == -1 and "65536" in str(e.value)
    assert wf.trace_closest_noalloc(np.zeros((0, 6), np.float32))["tri"].size == 0
    assert wf.stats()["stack_overflows"] == 0
    wf.close()


def test_noalloc_trace_allocates_nothing(bundled):
    """The property the entry point exists for: across several calls the library's count of device allocations
    (pt_stats.device_allocations: every hipMalloc of the library passes one wrapper) does not move and the device's free memory
    is the same to the byte, while pt_trace_closest moves the count with every call.  Also through a context over two
    members (routed to its first member)."""
    import torch
    import pathtracer_amd as pa
    mats, tris, nodes = bundled.arrays()
    rng = np.random.default_rng(29)
    for kw in ({}, {"device_count": 2, "group_flags": 1}):      # 1 = PT_GROUP_SAME_DEVICE
        wf = pa.Wavefront(**kw)
        wf.upload_scene(mats, tris, nodes)
        rays = _rays(rng, 4096)
        want = wf.trace_closest(rays)
        wf.trace_closest_noalloc(rays)                           # warm-up: code objects, the first use of the staging array
        torch.cuda.synchronize()
        n0, free0 = wf.stats()["device_allocations"], torch.cuda.mem_get_info()[0]
        for n in (4096, 65536, 17):
            got = wf.trace_closest_noalloc(rays[:n] if n <= 4096 else _rays(rng, n))
            if n <= 4096:
                assert np.array_equal(got["tri"], want["tri"][:n])
        torch.cuda.synchronize()
        n1, free1 = wf.stats()["device_allocations"], torch.cuda.mem_get_info()[0]
        assert n1 == n0, f"{n1 - n0} device allocations during pt_trace_closest_noalloc"
        assert free1 == free0
        wf.trace_closest(rays)
        assert wf.stats()["device_allocations"] > n1, "the counter must see pt_trace_closest's per-call buffers"
        wf.close()
