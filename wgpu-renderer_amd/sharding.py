"""Work partitioning for the three multi-GPU workloads.  The reference is single-device; these
partitions exist only where the work divides without changing any result:

  * orbit frames      - frame f of a camera path goes to rank f mod world (no exchange)
  * prefilter shards  - the levels that are cheaper to filter than to send (roughness 0: level 0, three
                        quarters of the map's bytes and a fiftieth of its work) are filtered by every rank
                        for itself; the texels of the other mips, flattened in (mip, face, y, x) order,
                        are cut into `world` ranges of equal cost (the library's own rule,
                        pbr_ibl_prefilter_partition) and only those are exchanged, with one all-gather
  * screen strips     - rank g owns an even-aligned band of rows; bands are gathered to rank 0

Everything here is host arithmetic (the prefilter cuts come from a host-only call of the library) so
it can be tested without a GPU.
"""


def orbit_frames(rank, world, n_frames):
    return list(range(rank, n_frames, world))


def prefilter_total_texels(size, mips):
    return sum(6 * max(1, size >> m) ** 2 for m in range(mips))


def prefilter_cuts(size, mips, world):
    """The library's own partition (pbr_ibl_prefilter_partition, host arithmetic, no device needed), world + 1
    numbers: cuts[0] texels at the front of the map are filtered by every shard for itself (0 for one shard), shard k
    owns texels [cuts[k], cuts[k + 1]), cuts[world] is the total.  A texel of level m weighs the GGX samples kept at
    that level's roughness, so the shards carry equal work, not equal texel counts."""
    import ctypes as C
    from . import _lib
    cuts = (C.c_uint64 * (world + 1))()
    _lib.check(None, _lib.load().pbr_ibl_prefilter_partition(size, mips, world, cuts))
    return [int(x) for x in cuts]


def prefilter_kept_samples(mips):
    """Samples of the 1024 that contribute at each level (1 at roughness 0)."""
    import ctypes as C
    from . import _lib
    out = []
    for level in range(mips):
        n = C.c_int32()
        _lib.check(None, _lib.load().pbr_ibl_prefilter_kept_samples(mips, level, C.byref(n)))
        out.append(n.value)
    return out


def prefilter_shard_texels(size, mips, rank, world):
    """[begin, end) in texels of what only rank `rank` filters, i.e. what it sends; identical to
    pbr_ibl_prefilter_shard_range / 16 bytes."""
    cuts = prefilter_cuts(size, mips, world)
    return cuts[rank], cuts[rank + 1]


def prefilter_replicated_texels(size, mips, world):
    """Texels at the front of the map that every rank filters for itself and nobody sends."""
    return prefilter_cuts(size, mips, world)[0]


def prefilter_exchange_bytes(size, mips, world, how="p2p"):
    """Bytes every rank contributes to the exchange of the shards (16 per texel): its own range when the ranges go
    peer to peer, the largest range when one all-gather of equal chunks carries them."""
    cuts = prefilter_cuts(size, mips, world)
    own = [16 * (e - b) for b, e in zip(cuts[:-1], cuts[1:])]
    return own if how == "p2p" else [max(own)] * world


def strip_rows(height, rank, world):
    """Even-aligned row band [y0, y1) of rank `rank`; quads (2x2 pixels) never straddle two bands."""
    pairs = (height + 1) // 2
    y0 = 2 * (pairs * rank // world)
    y1 = min(height, 2 * (pairs * (rank + 1) // world))
    return y0, y1


def prefilter_pieces(cuts, pieces):
    """The shards' ranges [cuts[k], cuts[k + 1]) cut once more into `pieces` consecutive parts each: entry j is the
    list of every rank's j-th part (first texel, end).  A rank filters and sends its parts one after the other, so
    that all but the last part travel while the next is being filtered (parallel.prefilter_sharded)."""
    pieces = max(1, int(pieces))
    return [[(b + (e - b) * j // pieces, b + (e - b) * (j + 1) // pieces) for b, e in zip(cuts[:-1], cuts[1:])] for j in range(pieces)]
