"""Work partitioning for the three multi-GPU workloads.  The reference is single-device; these
partitions exist only where the work divides without changing any result:

  * orbit frames      - frame f of a camera path goes to rank f mod world (no exchange)
  * prefilter shards  - the texels of all prefilter mips, flattened in (mip, face, y, x) order,
                        are cut into `world` equal ranges (same rule as pbr_ibl_prefilter); the
                        ranges are exchanged with one all-gather
  * screen strips     - rank g owns an even-aligned band of rows; bands are gathered to rank 0

Everything here is plain integer arithmetic so it can be tested without a GPU.
"""


def orbit_frames(rank, world, n_frames):
    return list(range(rank, n_frames, world))


def prefilter_total_texels(size, mips):
    return sum(6 * max(1, size >> m) ** 2 for m in range(mips))


def prefilter_shard_texels(size, mips, rank, world):
    """[begin, end) in texels; identical to pbr_ibl_prefilter_shard_range / 16 bytes."""
    total = prefilter_total_texels(size, mips)
    return total * rank // world, total * (rank + 1) // world


def prefilter_padded_chunk(size, mips, world):
    """All-gather needs equal chunks: the largest shard, in texels."""
    return max(e - b for b, e in (prefilter_shard_texels(size, mips, r, world) for r in range(world)))


def strip_rows(height, rank, world):
    """Even-aligned row band [y0, y1) of rank `rank`; quads (2x2 pixels) never straddle two bands."""
    pairs = (height + 1) // 2
    y0 = 2 * (pairs * rank // world)
    y1 = min(height, 2 * (pairs * (rank + 1) // world))
    return y0, y1
