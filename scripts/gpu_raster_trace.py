"""Per-tile timeline of the rasteriser's launch of one 4K frame, or with --stress of the 8K frame of 10 x 10 helmets
(diagnostic build, see gpu_surface_trace.py)."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry

pkg = entry.load_package()
stress = "--stress" in sys.argv
if stress:
    import bench
    g = pkg.glTFLoader(); g.load(pkg.renderer.DEFAULT_MODEL)
    r = pkg.Renderer(width=7680, height=4320, skybox_size=1024, instance_models=bench.stress_instances(pkg, g.get_geometries()[0].model))
    r.init()
    r.camera.update_camera_distance((5.0 - 30.0) * 100.0, 0.0)
else:
    r = pkg.Renderer(width=3840, height=2160, skybox_size=1024)
    r.init()
for _ in range(5):
    _, st = r.draw_frame(want_stats=True)
lib = pkg._lib.load()
raw = np.zeros((32768, 8), dtype=np.uint64)
fn = lib.pbr_debug_raster_trace
fn.argtypes = [C.c_void_p, C.c_size_t]
assert fn(raw.ctypes.data, raw.nbytes) == 0
np.save(os.path.join(ROOT, "gpurun_out", "raster_trace.npy"), raw)
tiles = 240 * 135 if stress else 120 * 68
t = raw[:tiles].astype(np.int64)
t0 = t[:, 0].min()
start, mid, end, n = (t[:, 0] - t0) * 0.01, (t[:, 1] - t0) * 0.01, (t[:, 2] - t0) * 0.01, t[:, 3]
geo = n > 0
out = {"ms_raster_event": float(st.ms_raster), "tiles": tiles, "tiles_with_geometry": int(geo.sum()), "entries": {"sum": int(n.sum()), "max": int(n.max()), "p90": float(np.percentile(n[geo], 90)), "median": float(np.median(n[geo]))},
       "start_us": {"median": float(np.median(start)), "max": float(start.max())}, "end_us": {"median": float(np.median(end)), "p99": float(np.percentile(end, 99)), "max": float(end.max())},
       "geometry_tiles": {"start_median": float(np.median(start[geo])), "start_max": float(start[geo].max()), "loop_us_median": float(np.median((mid - start)[geo])), "loop_us_p90": float(np.percentile((mid - start)[geo], 90)),
                          "loop_us_max": float((mid - start)[geo].max()), "end_max": float(end[geo].max())},
       "sky_tiles": {"dur_median": float(np.median((end - start)[~geo])), "dur_p90": float(np.percentile((end - start)[~geo], 90))}}
order = np.argsort(-(mid - start))[:8]
setup_done, kept, serve_done, blocks = (t[:, 4] - t0) * 0.01, t[:, 5], (t[:, 6] - t0) * 0.01, t[:, 7]
out["slowest_loops"] = [{"tile": int(i), "entries": int(n[i]), "start": float(start[i]), "loop_us": float(mid[i] - start[i]), "end": float(end[i]), "wave0_setup_us": float(setup_done[i] - start[i]),
                         "wave0_kept": int(kept[i]), "wave0_serve_us": float(serve_done[i] - setup_done[i]), "wave0_blocks": int(blocks[i])} for i in order]
med = np.argsort(np.abs(n - 18) + (~geo) * 1000)[:6]
out["typical_tiles"] = [{"tile": int(i), "entries": int(n[i]), "start": float(start[i]), "loop_us": float(mid[i] - start[i]), "wave0_setup_us": float(setup_done[i] - start[i]), "wave0_kept": int(kept[i]),
                         "wave0_serve_us": float(serve_done[i] - setup_done[i]), "wave0_blocks": int(blocks[i])} for i in med]
step = 10 if stress else 1
busy = [(int(e), int(((start <= e) & (end > e)).sum())) for e in range(0, int(end.max()) + 2, step)]
if stress:
    dur = end - start
    out["by_entries"] = [{"entries": [lo, hi], "tiles": int(sel.sum()), "loop_us_median": float(np.median((mid - start)[sel])), "dur_us_median": float(np.median(dur[sel])), "dur_us_sum": float(dur[sel].sum())}
                         for lo, hi in ((0, 0), (1, 31), (32, 63), (64, 127), (128, 255), (256, 511), (512, 1 << 20)) for sel in [(n >= lo) & (n <= hi)] if sel.any()]
out["busy_tiles_by_us"] = busy
print(json.dumps(out))
