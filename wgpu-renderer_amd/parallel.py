"""Multi-GPU drivers on torch.distributed (backend "nccl" = RCCL over xGMI on MI355X, "gloo" on
CPU for tests).  One process per GPU.  Collectives appear only where the partition has a real
exchange step: the all-gather of prefilter shards and the gather of screen strips."""
import ctypes as C

import torch
import torch.distributed as dist

from . import _lib, sharding


class _DevMem:
    """Expose a raw device range as a torch tensor through __cuda_array_interface__."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def device_bytes_tensor(ptr, nbytes, device):
    return torch.as_tensor(_DevMem(ptr, nbytes), device=device)


def prefilter_sharded(renderer, size, mips, rank=None, world=None, device=None):
    """Each rank filters its texel range, then one all-gather makes the whole map resident on every
    GPU (so every GPU can shade).  Returns nothing; the map lives in the renderer's context."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    lib = renderer._lib
    _lib.check(renderer._ctx, lib.pbr_ibl_prefilter(renderer._ctx, size, mips, 1024, rank, world))
    if world == 1:
        return
    ptr, total = C.c_void_p(), C.c_size_t()
    _lib.check(renderer._ctx, lib.pbr_ibl_device_ptr(renderer._ctx, _lib.MAP_PREFILTER, C.byref(ptr), C.byref(total)))
    device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    whole = device_bytes_tensor(ptr.value, total.value, device)
    renderer.synchronize()
    all_gather_ranges(whole, [tuple(16 * t for t in sharding.prefilter_shard_texels(size, mips, r, world)) for r in range(world)], rank)
    # the gather ran on torch's stream and wrote through the library's pointer: finish it before the
    # renderer's stream reads the map (the library rebuilds its derived copy at the next frame, which
    # the pbr_ibl_device_ptr call above has requested)
    if whole.is_cuda:
        torch.cuda.current_stream(whole.device).synchronize()


def all_gather_ranges(whole, byte_ranges, rank):
    """In-place all-gather of unequal contiguous byte ranges of `whole` (a 1-D uint8 tensor that
    every rank holds, each with its own range filled in)."""
    world = len(byte_ranges)
    chunk = max(e - b for b, e in byte_ranges)
    send = torch.zeros(chunk, dtype=torch.uint8, device=whole.device)
    b, e = byte_ranges[rank]
    send[:e - b] = whole[b:e]
    recv = torch.empty(chunk * world, dtype=torch.uint8, device=whole.device)
    dist.all_gather_into_tensor(recv, send)
    for r, (rb, re) in enumerate(byte_ranges):
        if r != rank:
            whole[rb:re] = recv[r * chunk:r * chunk + (re - rb)]


def gather_strips(strip, height, width, rank=None, world=None, dst=0):
    """Gather each rank's [rows, width, 4] uint8 band to rank `dst`; returns the full frame there."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    rows = [sharding.strip_rows(height, r, world) for r in range(world)]
    most = max(y1 - y0 for y0, y1 in rows)
    send = torch.zeros((most, width, 4), dtype=torch.uint8, device=strip.device)
    send[:strip.shape[0]] = strip
    bufs = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, bufs, dst=dst)
    if rank != dst:
        return None
    frame = torch.empty((height, width, 4), dtype=torch.uint8, device=strip.device)
    for r, (y0, y1) in enumerate(rows):
        frame[y0:y1] = bufs[r][:y1 - y0]
    return frame
