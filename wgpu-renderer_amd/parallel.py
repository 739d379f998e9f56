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


def prefilter_sharded(renderer, size, mips, rank=None, world=None, device=None, how="p2p", stream=None, comm_stream=None, rehearse=False, pieces=2):
    """Each rank filters its own range of the levels that are worth sharding, then the ranges are exchanged so that the
    whole map is resident on every GPU (every GPU can shade), and the levels that are cheaper to filter than to send
    (level 0) are filtered by every rank for itself.  With `stream` - the torch stream the renderer's context runs on
    (Renderer.set_stream) - the exchange runs on a second stream beside the filtering (`comm_stream`: the second
    stream, else one is created and kept): the rank's range is filtered in `pieces` parts, each part travels while the
    next is filtered, and the last one while level 0 is; without `stream` the steps follow each other.  `rehearse`
    takes a single rank through the same steps (its range is then the whole map and the exchange has no peer).
    Returns the bytes this rank contributed to the exchange; the map lives in the renderer's context."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    lib = renderer._lib
    ctx = renderer._ctx
    if world == 1 and not rehearse:
        _lib.check(ctx, lib.pbr_ibl_prefilter(ctx, size, mips, 1024, 0, 1))
        return 0
    cuts = sharding.prefilter_cuts(size, mips, world)
    ptr, total = C.c_void_p(), C.c_size_t()
    whole, comm, exchanged, sent = None, None, None, 0
    for part in sharding.prefilter_pieces(cuts, pieces):
        first, end = part[rank]
        # this rank's part (the first call also allocates the map when its size changed) ...
        _lib.check(ctx, lib.pbr_ibl_prefilter_texels(ctx, size, mips, 1024, first, end - first))
        if whole is None:
            _lib.check(ctx, lib.pbr_ibl_device_ptr(ctx, _lib.MAP_PREFILTER, C.byref(ptr), C.byref(total)))
            device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
            whole = device_bytes_tensor(ptr.value, total.value, device)
        ranges = [(16 * b, 16 * e) for b, e in part]
        if stream is not None and whole.is_cuda:
            # ... and its exchange on a second stream, behind an event, while the next part is filtered
            filtered = torch.cuda.Event()
            filtered.record(stream)
            comm = comm if comm is not None else comm_stream if comm_stream is not None else _comm_stream(whole.device)
            comm.wait_event(filtered)
            with torch.cuda.stream(comm):
                sent += all_gather_ranges(whole, ranges, rank, how=how)
                exchanged = torch.cuda.Event()
                exchanged.record(comm)
        else:
            renderer.synchronize()
            sent += all_gather_ranges(whole, ranges, rank, how=how)
            if whole.is_cuda:
                torch.cuda.current_stream(whole.device).synchronize()
    # meanwhile the leading texels that nobody sends
    if cuts[0]:
        _lib.check(ctx, lib.pbr_ibl_prefilter_texels(ctx, size, mips, 1024, 0, cuts[0]))
    if exchanged is not None:
        stream.wait_event(exchanged)     # the renderer's next work on the map follows the exchange
    # the peers wrote through the library's pointer: ask for the map's derived copy to be rebuilt at the next frame
    _lib.check(ctx, lib.pbr_ibl_device_ptr(ctx, _lib.MAP_PREFILTER, C.byref(ptr), C.byref(total)))
    return sent


_comm_streams = {}


def _comm_stream(device):
    """One stream per device for the exchanges that overlap the renderer's work; created once."""
    key = str(device)
    if key not in _comm_streams:
        _comm_streams[key] = torch.cuda.Stream(device)
    return _comm_streams[key]


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
