"""Host time to issue a 4K frame (uniform block ready, three launches through the C ABI) against the GPU time per
frame: python scripts/gpu_host_issue_rate.py [frames].  The host must stay ahead of the GPU for the frames in
flight to overlap."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as entry
pkg = entry.load_package()
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 300
lanes = []
for _ in range(3):
    r = pkg.Renderer(width=3840, height=2160, skybox_size=1024)
    r.init()
    s = torch.cuda.Stream()
    r.set_stream(s.cuda_stream)
    lanes.append((r, torch.empty((2160, 3840), dtype=torch.int32, device="cuda:0")))
u = lanes[0][0].frame_uniforms()
for i in range(30):
    lanes[i % 3][0].draw_frame_device(lanes[i % 3][1].data_ptr(), uniforms=u)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(frames):
    lanes[i % 3][0].draw_frame_device(lanes[i % 3][1].data_ptr(), uniforms=u)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host issue {(t1 - t0) / frames * 1e6:.1f} us per frame, all frames done after {(t2 - t0) / frames * 1e6:.1f} us per frame", flush=True)
