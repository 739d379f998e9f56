"""The two exchanges of the multi-GPU drivers with more than one rank on the one GPU of the test box (gloo, since two ranks
cannot share one GPU under RCCL; the data crosses on the host).

The screen-strip split across ranks: two processes (both on the one GPU of the test box) each upload the instanced
scene, render their even-aligned band through the C ABI and gather the bands to rank 0 with the same
parallel.gather_strips the benchmark's strips mode uses (gloo here, since two ranks cannot share one GPU under RCCL):
rank 0's gathered frame must equal the frame one process renders alone, byte for byte."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, GRID = 800, 450, 3


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _scene(pkg):
    loader = pkg.glTFLoader(); loader.load(pkg.renderer.DEFAULT_MODEL)
    base = np.asarray(loader.get_geometries()[0].model, dtype=np.float32)
    mx = pkg.mathx
    models = np.stack([mx.mul(mx.translate(mx.identity(), [(i - 1) * 2.5, (j - 1) * 2.5, 0.0]), base) for j in range(GRID) for i in range(GRID)])
    r = pkg.Renderer(width=W, height=H, hdr=pkg.synthetic_sky(1024, 512), skybox_size=64, irradiance_size=8, prefilter_size=32, prefilter_mips=4, lut_size=64,
                     instance_models=models)
    r.init()
    r.camera.update_camera_distance(-700.0, 0.0)
    r.camera.update_camera_rotation(-200.0, 100.0)
    return r


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as entry
    pkg = entry.load_package()
    from wgpu_renderer_amd import parallel, sharding
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        r = _scene(pkg)
        y0, y1 = sharding.strip_rows(H, rank, world)
        strip = torch.empty((y1 - y0, W, 4), dtype=torch.uint8, device="cuda:0")
        r.draw_frame_device(strip.data_ptr(), y0=y0, y1=y1)
        r.synchronize()
        frame = parallel.gather_strips(strip.cpu(), H, W, rank, world, dst=0)
        if rank == 0:
            whole = r.draw_frame()
            q.put((rank, bool((frame.numpy() == whole).all()), int((whole[..., :3].max(-1) > 0).sum())))
        else:
            q.put((rank, frame is None, 0))
        r.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gathered_strips_equal_the_single_process_frame(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, ok, lit in results:
        assert ok, (rank, ok)
        if rank == 0:
            assert lit > W * H // 2


def _prefilter_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import ctypes as C
    import hashlib
    import torch
    import torch.distributed as dist
    import __graft_entry__ as entry
    pkg = entry.load_package()
    lib = pkg._lib
    from wgpu_renderer_amd import parallel, sharding
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        size, mips = 128, 5
        hdr = pkg.synthetic_sky(2048, 1024)
        r = pkg.Renderer(width=64, height=64, hdr=hdr)
        r._open()
        data = np.ascontiguousarray(hdr.data, dtype=np.float32)
        lib.check(r._ctx, r._lib.pbr_upload_hdr(r._ctx, data.ctypes.data_as(C.c_void_p), hdr.width, hdr.height))
        # what parallel.prefilter_sharded does on the device: the rank's own range, and the leading texels nobody sends
        cuts = sharding.prefilter_cuts(size, mips, world)
        lib.check(r._ctx, r._lib.pbr_ibl_prefilter_texels(r._ctx, size, mips, 1024, cuts[rank], cuts[rank + 1] - cuts[rank]))
        lib.check(r._ctx, r._lib.pbr_ibl_prefilter_texels(r._ctx, size, mips, 1024, 0, cuts[0]))
        r.synchronize()
        mine = np.concatenate([r.download_map(lib.MAP_PREFILTER, m).reshape(-1) for m in range(mips)])
        whole = torch.from_numpy(mine.view(np.uint8).copy())
        ranges = [(16 * b, 16 * e) for b, e in zip(cuts[:-1], cuts[1:])]
        # a rank holds nothing of the other ranks' ranges before the exchange
        other = [whole[b:e] for k, (b, e) in enumerate(ranges) if k != rank and e > b]
        nothing_yet = all(not bool(t.any()) for t in other)
        sent = 0
        for part in sharding.prefilter_pieces(cuts, 2):      # in two parts per rank, as prefilter_sharded sends them
            sent += parallel.all_gather_ranges(whole, [(16 * b, 16 * e) for b, e in part], rank, how="p2p")
        lib.check(r._ctx, r._lib.pbr_ibl_prefilter(r._ctx, size, mips, 1024, 0, 1))      # the whole map, by this process alone
        alone = np.concatenate([r.download_map(lib.MAP_PREFILTER, m).reshape(-1) for m in range(mips)]).view(np.uint8)
        q.put((rank, nothing_yet, bool((whole.numpy() == alone).all()), sent == ranges[rank][1] - ranges[rank][0], cuts[0] == 6 * size * size))
        r.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_prefilter_equals_the_single_process_map(world):
    """Every rank filters its own range of levels 1.. and all of level 0, the ranges are exchanged peer to peer, and
    every rank then holds the bytes one process computes alone."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_prefilter_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for res in results:
        assert all(res[1:]), res
