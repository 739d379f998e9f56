"""Times the 4K frame in the single-sample and the 4x multisampled mode: kernel by kernel from the frame
statistics (which make every tile's visibility entries be written), and the stream time per frame of a run
of colour-only frames.  python scripts/gpu_msaa_frame.py [frames]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as entry
pkg = entry.load_package()
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 20
r = pkg.Renderer(width=3840, height=2160, skybox_size=1024)
r.init()
color = torch.empty((2160, 3840), dtype=torch.int32, device="cuda:0")
for msaa in (False, True):
    r.msaa_options["enable"] = msaa
    rows = []
    for _ in range(frames):
        _, st = r.draw_frame(want_stats=True)
        rows.append((st.ms_vertex, st.ms_setup, st.ms_raster, st.ms_resolve, st.ms_total))
    m = np.median(np.array(rows[3:]), axis=0) * 1e3
    print(f"msaa={int(msaa)} us: vertex {m[0]:.1f} setup {m[1]:.1f} raster {m[2]:.1f} resolve {m[3]:.1f} total {m[4]:.1f} covered {st.covered_pixels}", flush=True)
    u = r.frame_uniforms()
    for _ in range(5):
        r.draw_frame_device(color.data_ptr(), uniforms=u)
    r.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        r.draw_frame_device(color.data_ptr(), uniforms=u)
    r.synchronize()
    print(f"msaa={int(msaa)} colour-only frames: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per frame", flush=True)
