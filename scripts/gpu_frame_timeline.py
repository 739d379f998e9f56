"""Per-kernel event times of single 4K frames (one at a time, nothing else on the chip): python scripts/gpu_frame_timeline.py [frames]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as entry
pkg = entry.load_package()
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 20
r = pkg.Renderer(width=3840, height=2160, skybox_size=1024)
r.init()
out = torch.empty((2160, 3840, 4), dtype=torch.uint8, device="cuda:0")
st = pkg._lib.FrameStats()
acc = [0.0] * 4
for i in range(frames + 5):
    r.draw_frame_device(out.data_ptr(), stats=st)
    if i >= 5:
        acc = [a + b for a, b in zip(acc, (st.ms_setup, st.ms_raster, st.ms_resolve, st.ms_total))]
    r.camera.update_camera_rotation(-10.0, 0.0)
print("us per frame: setup %.1f raster %.1f surface %.1f total %.1f | quads uniform %d mixed %d covered %d tris %d spill refs %d" % tuple(
    [1e3 * a / frames for a in acc] + [st.uniform_quads, st.mixed_quads, st.covered_pixels, st.setup_triangles, st.binned_refs]))
