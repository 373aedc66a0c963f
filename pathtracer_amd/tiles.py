"""Image tiles across the GPUs of one node: one process per GPU, scene replicated, tiles sharded.

Pixel group g (an 8x8 Morton tile, generate.glsl:177-187) is rendered by rank g % world into
that rank's own accumulator.  Nothing is exchanged while rendering; once per image the
accumulators are gathered to rank 0 (RCCL over xGMI when the process group's backend is
"nccl": a direct gather, each rank sends W*H*16/world bytes once) and de-interleaved there by
pt_assemble_tiles.  torch is used only for the process group and as a view of device memory.
"""
import numpy as np


def tile_owner(width, height, world):
    """rank owning each pixel, shape (height, width) — the partition the kernels implement."""
    gcx = (width + 7) // 8
    ys, xs = np.mgrid[0:height, 0:width]
    group = (ys // 8) * gcx + (xs // 8)
    return (group % world).astype(np.int32)


def local_slots(width, height, rank, world):
    """(own slot count, padded slot count) of a rank; the padded count is equal on every rank."""
    groups = ((width + 7) // 8) * ((height + 7) // 8)
    return ((groups + world - 1 - rank) // world) * 64, ((groups + world - 1) // world) * 64


class _DevView:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n, 4), "typestr": "<f4", "data": (ptr, False), "version": 3, "strides": None}


def accumulator_tensor(wf):
    """Zero-copy torch view (n_slots_padded, 4) of a Wavefront's device accumulator."""
    import torch
    ptr, n, _ = wf.accum_device()
    return torch.as_tensor(_DevView(ptr, int(n)), device=torch.device("cuda", torch.cuda.current_device()))


def gather_accumulators(local, dst=0, group=None):
    """Gather every rank's accumulator tensor to `dst`.  Returns (world, n, 4) there, None elsewhere.

    With the "nccl" backend the tensors stay on the GPU and move over xGMI via RCCL.  Other backends
    (gloo: the CPU tests and single-GPU rehearsals of the multi-rank path) carry CPU copies; the result
    comes back on the device `local` lives on.
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return local.unsqueeze(0)
    device = local.device
    staged = local
    if dist.get_backend(group) != "nccl" and local.is_cuda:
        staged = local.cpu()
    if rank == dst:
        out = torch.empty((world,) + tuple(staged.shape), dtype=staged.dtype, device=staged.device)
        dist.gather(staged, list(out.unbind(0)), dst=dst, group=group)
        return out.to(device)
    dist.gather(staged, None, dst=dst, group=group)
    return None


class AccumulatorGather:
    """Once-per-image gather of one context's accumulator with preallocated buffers.

    Each call snapshots the accumulator and gathers the snapshot.  With the "nccl" backend nothing blocks the host: the
    snapshot is a device copy queued on the context's OWN stream (after this image's kernels and before whatever the
    caller enqueues next — the next pt_render or pt_reset_accumulation cannot overtake it), torch's current stream waits
    for it through an event, and RCCL reads the snapshot while the context already renders the next image into the live
    accumulator.  The next call's snapshot waits, again through an event, for this call's gather to have finished
    reading the buffer.  `result` is the (world, n, 4) tensor on `dst` (None elsewhere); it is complete once torch's
    current stream has caught up (torch.cuda.synchronize()).
    """

    def __init__(self, wf, dst=0, group=None):
        import torch
        import torch.distributed as dist
        self.wf, self.dst, self.group = wf, dst, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.on_device = dist.get_backend(group) == "nccl"
        self.local = accumulator_tensor(wf)
        dev = self.local.device if self.on_device else torch.device("cpu")
        self.snapshot = torch.empty(self.local.shape, dtype=self.local.dtype, device=dev)
        self.result = None
        if self.rank == dst:
            self.result = torch.empty((self.world,) + tuple(self.local.shape), dtype=self.local.dtype, device=dev)
        self._ctx_stream = torch.cuda.ExternalStream(wf.stream(), device=self.local.device) if self.on_device else None
        self._gathered = None   # event: the previous gather has finished with the snapshot buffer

    def __call__(self):
        import torch
        import torch.distributed as dist
        parts = list(self.result.unbind(0)) if self.rank == self.dst else None
        if self.on_device:
            mine, cur = self._ctx_stream, torch.cuda.current_stream()
            if self._gathered is not None:
                mine.wait_event(self._gathered)
            with torch.cuda.stream(mine):
                self.snapshot.copy_(self.local, non_blocking=True)
            cur.wait_event(mine.record_event())
            dist.gather(self.snapshot, parts, dst=self.dst, group=self.group)
            self._gathered = cur.record_event()
        else:
            self.wf.synchronize()
            self.snapshot.copy_(self.local)
            dist.gather(self.snapshot, parts, dst=self.dst, group=self.group)
        return self.result


def gather_image(wf, samples, dst=0, group=None):
    """Synchronise the context, gather all ranks' tiles and assemble the image on `dst` (None elsewhere)."""
    import torch
    import torch.distributed as dist
    wf.synchronize()
    local = accumulator_tensor(wf)
    gathered = gather_accumulators(local, dst=dst, group=group)
    if gathered is None:
        return None
    torch.cuda.synchronize()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    return wf.assemble_tiles(gathered.data_ptr(), world, samples)
