"""Throughput of the orbit with one context vs two contexts on one GPU (frames alternate between them,
each context on its own stream, so the latency-bound geometry kernels of one frame overlap the
shading of the other)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as entry
pkg = entry.load_package()
W, H, N = 3840, 2160, 100
dev = torch.device("cuda", 0)
def make():
    r = pkg.Renderer(width=W, height=H, skybox_size=1024); r.init()
    s = torch.cuda.Stream(dev); r.set_stream(s.cuda_stream)
    c = torch.empty((H, W, 4), dtype=torch.uint8, device=dev)
    return r, s, c
ctxs = [make() for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2)]
for n_ctx in range(1, len(ctxs) + 1):
    use = ctxs[:n_ctx]
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for f in range(N):
            r, s, c = use[f % n_ctx]
            r.draw_frame_device(c.data_ptr())
            r.camera.update_camera_rotation(-10.0 * n_ctx, 0.0)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"{n_ctx} context(s): {dt / N * 1e6:.1f} us per frame, {N * W * H / dt / 1e9:.1f} Gpix/s", flush=True)
