"""Resolve/raster time versus coverage at 4K: fits the per-pixel cost of sky and surface shading."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package()
W, H = int(os.environ.get("W", 3840)), int(os.environ.get("H", 2160))
r = pkg.Renderer(width=W, height=H, skybox_size=1024)
r.init()
for dist in (5.0, 3.5, 2.5, 1.8, 60.0):
    r.camera = pkg.Camera()
    r.camera.update_camera_distance((5.0 - dist) * 100.0, 0.0)
    best = None
    for _ in range(5):
        _, st = r.draw_frame(want_stats=True)
        if best is None or st.ms_total < best[0]:
            best = (st.ms_total, st.ms_vertex, st.ms_setup, st.ms_raster, st.ms_resolve, st.covered_pixels, st.setup_triangles, st.binned_refs, st.uniform_quads, st.mixed_quads)
    c = best[5] / (W * H)
    print(f"dist {dist:5.1f} cover {c:.4f} total {best[0]*1e3:7.1f} us  vertex {best[1]*1e3:6.1f} setup+bin {best[2]*1e3:6.1f} raster {best[3]*1e3:6.1f} resolve {best[4]*1e3:6.1f}  tris {best[6]} refs {best[7]} quads u {best[8]} m {best[9]}", flush=True)
