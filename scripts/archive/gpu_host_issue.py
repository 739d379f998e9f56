"""How long the host takes to issue the benchmark's frames (three launches each, through ctypes) against how long the
GPU takes to draw them: the timed loop of bench.py with the clock read before and after the final synchronisation."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as entry

pkg = entry.load_package()
K, N = int(sys.argv[1]) if len(sys.argv) > 1 else 4, 360
dev = torch.device("cuda:0")
r = pkg.Renderer(width=3840, height=2160, skybox_size=1024)
r.init()
streams = [torch.cuda.Stream(dev) for _ in range(K)]
lanes = []
for k in range(K):
    rr = r if k == 0 else r.fork()
    rr.set_stream(streams[k].cuda_stream)
    lanes.append((rr, torch.empty((2160, 3840, 4), dtype=torch.uint8, device=dev)))
uniforms = []
for _ in range(N + 20):
    uniforms.append(r.frame_uniforms())
    r.camera.update_camera_rotation(-10.0, 0.0)
for i in range(20):
    lanes[i % K][0].draw_frame_device(lanes[i % K][1].data_ptr(), uniforms=uniforms[i])
torch.cuda.synchronize(dev)
out = {}
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(N):
        lanes[i % K][0].draw_frame_device(lanes[i % K][1].data_ptr(), uniforms=uniforms[20 + i])
    t1 = time.perf_counter()
    torch.cuda.synchronize(dev)
    t2 = time.perf_counter()
    out[f"run{rep}"] = {"host_issue_us_per_frame": (t1 - t0) / N * 1e6, "total_us_per_frame": (t2 - t0) / N * 1e6}
print(json.dumps(out))
for rr, _ in lanes[1:]:
    rr.close()
