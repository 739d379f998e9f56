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


def prefilter_sharded(renderer, size, mips, rank=None, world=None, device=None, how="p2p"):
    """Each rank filters the levels that are cheaper to filter than to send (level 0) plus its own range of the
    others, then one exchange of those ranges makes the whole map resident on every GPU (so every GPU can shade).
    Returns the bytes this rank contributed to the exchange; the map lives in the renderer's context."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    lib = renderer._lib
    _lib.check(renderer._ctx, lib.pbr_ibl_prefilter(renderer._ctx, size, mips, 1024, rank, world))
    if world == 1:
        return 0
    ptr, total = C.c_void_p(), C.c_size_t()
    _lib.check(renderer._ctx, lib.pbr_ibl_device_ptr(renderer._ctx, _lib.MAP_PREFILTER, C.byref(ptr), C.byref(total)))
    device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    whole = device_bytes_tensor(ptr.value, total.value, device)
    renderer.synchronize()
    sent = all_gather_ranges(whole, [tuple(16 * t for t in sharding.prefilter_shard_texels(size, mips, r, world)) for r in range(world)], rank, how=how)
    # the gather ran on torch's stream and wrote through the library's pointer: finish it before the
    # renderer's stream reads the map (the library rebuilds its derived copy at the next frame, which
    # the pbr_ibl_device_ptr call above has requested)
    if whole.is_cuda:
        torch.cuda.current_stream(whole.device).synchronize()
    return sent


_staging = {}


def _staging_for(device, send_bytes, recv_bytes):
    """Send and receive buffers of the equal-chunk collectives, allocated once per device and size and kept: the
    exchanges run every frame (strips) or every map (prefilter) with the same sizes."""
    key = (str(device), send_bytes, recv_bytes)
    if key not in _staging:
        _staging[key] = (torch.zeros(send_bytes, dtype=torch.uint8, device=device),
                         torch.empty(recv_bytes, dtype=torch.uint8, device=device) if recv_bytes else None)
    return _staging[key]


def release_staging():
    _staging.clear()


def all_gather_ranges(whole, byte_ranges, rank, how="p2p"):
    """In-place all-gather of unequal contiguous byte ranges of `whole` (a 1-D uint8 tensor that
    every rank holds, each with its own range filled in).  Returns the bytes this rank contributed.

    how="p2p" (the default): every rank sends its range straight from `whole` into every peer's `whole`, one batch of
    isend / irecv.  The GPUs of an MI355X node are fully connected, seven direct xGMI links each, so a rank's seven
    sends each have a link of their own; nothing is padded, staged or copied.
    how="collective": one all_gather_into_tensor of chunks as large as the largest range.  The ranges are cuts of
    equal cost, not of equal size (a third above the mean at eight ranks of the 2048^2 x 10 map), and pass through
    staging buffers that are allocated once and kept."""
    world = len(byte_ranges)
    b, e = byte_ranges[rank]
    if how == "p2p":
        ops = []
        for r, (rb, re) in enumerate(byte_ranges):
            if r == rank:
                continue
            if e > b:
                ops.append(dist.P2POp(dist.isend, whole[b:e], r))
            if re > rb:
                ops.append(dist.P2POp(dist.irecv, whole[rb:re], r))
        for req in (dist.batch_isend_irecv(ops) if ops else []):
            req.wait()
        return e - b
    chunk = max(re - rb for rb, re in byte_ranges)
    if chunk == 0:
        return 0
    send, recv = _staging_for(whole.device, chunk, chunk * world)
    send[:e - b] = whole[b:e]
    dist.all_gather_into_tensor(recv, send)
    for r, (rb, re) in enumerate(byte_ranges):
        if r != rank:
            whole[rb:re] = recv[r * chunk:r * chunk + (re - rb)]
    return chunk


def gather_strips(strip, height, width, rank=None, world=None, dst=0, out=None):
    """Gather each rank's [rows, width, 4] uint8 band to rank `dst`; returns the full frame there (in `out` when
    given).  Bands of the same height (the usual case) are received straight into the frame's rows; otherwise they
    pass through buffers of the tallest band's size, allocated once."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    rows = [sharding.strip_rows(height, r, world) for r in range(world)]
    most = max(y1 - y0 for y0, y1 in rows)
    if all(y1 - y0 == most for y0, y1 in rows):
        if rank != dst:
            dist.gather(strip, None, dst=dst)
            return None
        frame = out if out is not None else torch.empty((height, width, 4), dtype=torch.uint8, device=strip.device)
        dist.gather(strip, [frame[y0:y1] for y0, y1 in rows], dst=dst)
        return frame
    band = most * width * 4
    send, recv = _staging_for(strip.device, band, band * world if rank == dst else 0)
    send[:strip.numel()] = strip.reshape(-1)
    dist.gather(send, [recv[r * band:(r + 1) * band] for r in range(world)] if rank == dst else None, dst=dst)
    if rank != dst:
        return None
    frame = out if out is not None else torch.empty((height, width, 4), dtype=torch.uint8, device=strip.device)
    for r, (y0, y1) in enumerate(rows):
        frame[y0:y1] = recv[r * band:r * band + (y1 - y0) * width * 4].view(y1 - y0, width, 4)
    return frame
