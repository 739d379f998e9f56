"""The 8K stress configuration on one GPU: 10x10 instanced DamagedHelmet (spacing 2.5, camera distance 30) at
7680x4320, whole frame and one of eight 540-row strips."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package()
W, H = 7680, 4320
loader = pkg.glTFLoader(); loader.load(pkg.renderer.DEFAULT_MODEL)
base = np.asarray(loader.get_geometries()[0].model, dtype=np.float32)
models = []
for j in range(10):
    for i in range(10):
        t = pkg.mathx.translate(pkg.mathx.identity(), [(i - 4.5) * 2.5, (j - 4.5) * 2.5, 0.0])
        models.append(pkg.mathx.mul(t, base))
r = pkg.Renderer(width=W, height=H, skybox_size=1024, instance_models=np.stack(models))
r.init()
r.camera.update_camera_distance((5.0 - 30.0) * 100.0, 0.0)
for label, (y0, y1) in (("full frame", (0, H)), ("strip 3 of 8", (3 * 540, 4 * 540))):
    best = None
    for _ in range(4):
        out = r.draw_frame(y0=y0, y1=y1, want_stats=True)
        st = out[-1]
        if best is None or st.ms_total < best.ms_total:
            best = st
            rgba = out[0]
    print(f"{label}: total {best.ms_total*1e3:.0f} us vertex {best.ms_vertex*1e3:.0f} setup {best.ms_setup*1e3:.0f} raster {best.ms_raster*1e3:.0f} "
          f"resolve {best.ms_resolve*1e3:.0f}; covered {best.covered_pixels} tris {best.setup_triangles} spilled {best.binned_refs} quads u {best.uniform_quads} m {best.mixed_quads}; "
          f"mean rgb {rgba[..., :3].mean():.2f}", flush=True)
