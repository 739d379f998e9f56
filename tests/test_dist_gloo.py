"""World-size-2 checks of the multi-GPU host logic on CPU (gloo): the collectives used for the
prefilter all-gather and the strip gather, and the frame partition of the orbit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    pkg = entry.load_package()
    from wgpu_renderer_amd import parallel, sharding
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # 1. prefilter shards: every rank fills the part every rank computes and its own byte range of the rest, the
        #    exchange completes the map (both ways of exchanging; the staging of the collective one is reused)
        ok1 = True
        for size, mips in ((16, 3), (32, 5)):
            total = sharding.prefilter_total_texels(size, mips) * 16
            ranges = [tuple(16 * t for t in sharding.prefilter_shard_texels(size, mips, r, world)) for r in range(world)]
            replicated = 16 * sharding.prefilter_replicated_texels(size, mips, world)
            ok1 = ok1 and replicated == 16 * 6 * size * size == ranges[0][0] and ranges[-1][1] == total
            truth = (torch.arange(total, dtype=torch.int64) * 7 % 251).to(torch.uint8)
            for how in ("collective", "p2p", "collective"):
                whole = torch.zeros(total, dtype=torch.uint8)
                whole[:replicated] = truth[:replicated]
                b, e = ranges[rank]
                whole[b:e] = truth[b:e]
                sent = parallel.all_gather_ranges(whole, ranges, rank, how=how)
                ok1 = ok1 and bool((whole == truth).all())
                ok1 = ok1 and sent == (e - b if how == "p2p" else max(y - x for x, y in ranges))
        ok1 = ok1 and len(parallel._staging) == 2
        # 2. strips: even-aligned bands gathered to rank 0
        H, W = 46, 10
        frame = (torch.arange(H * W * 4, dtype=torch.int64) % 253).to(torch.uint8).reshape(H, W, 4)
        y0, y1 = sharding.strip_rows(H, rank, world)
        got = parallel.gather_strips(frame[y0:y1].clone(), H, W, rank, world, dst=0)
        ok2 = bool((got == frame).all()) if rank == 0 else got is None
        # bands of one height go straight into the rows of a frame the caller keeps
        H2 = 8 * world
        frame2 = (torch.arange(H2 * W * 4, dtype=torch.int64) % 241).to(torch.uint8).reshape(H2, W, 4)
        y0, y1 = sharding.strip_rows(H2, rank, world)
        out = torch.zeros_like(frame2) if rank == 0 else None
        got2 = parallel.gather_strips(frame2[y0:y1].clone(), H2, W, rank, world, dst=0, out=out)
        ok2 = ok2 and ((got2 is out and bool((out == frame2).all())) if rank == 0 else got2 is None)
        # 3. orbit frames: disjoint cover
        mine = sharding.orbit_frames(rank, world, 11)
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        ok3 = sorted(sum(gathered, [])) == list(range(11))
        # 4. max-over-ranks timing reduction used by bench.py
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ok4 = float(t.item()) == float(world)
        q.put((rank, ok1, ok2, ok3, ok4))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_collectives_world(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in results:
        assert all(r[1:]), r
