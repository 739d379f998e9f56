"""The screen-strip split across ranks: two processes (both on the one GPU of the test box) each upload the instanced
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
